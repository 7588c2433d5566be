// host_path.cpp — what the drop-in costs on the HOST side of the seam (src/receiver.cpp:140-175), stage by stage, for one cloud:
//   host cloud -> gndt_build (H2D + build) + gndt_sync -> gndt_export (D2H) -> the reference's containers (materialise_into)
//   -> computeCost (GPU flood + h into the Slope objects) -> AstarPlanar::findRoute
// in the eager mode (containers rebuilt: what code that walks map_cell / map_xy itself needs) and in the lazy mode (consumers
// served from the exported rows).  Built and run by bench.py ("host_path" block) and tools/measure_host_path.py.
//   host_path <cloud.f32> <n> <grid_len> <z_len> <interval> <demand> <gx> <gy> <gz> <sx> <sy> <sz> <radius> [reps]
// prints ONE JSON object (medians over reps, milliseconds).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gndt_compat.hpp"

using namespace gndt_compat;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
    if (argc < 14) { std::printf("usage: host_path cloud.f32 n grid_len z_len interval demand gx gy gz sx sy sz radius [reps]\n"); return 2; }
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    const float gl = std::strtof(argv[3], nullptr), zl = std::strtof(argv[4], nullptr), iv = std::strtof(argv[5], nullptr);
    const std::string demand = argv[6];
    Vector3f goal, start;
    for (int k = 0; k < 3; ++k) { goal.d[k] = std::strtof(argv[7 + k], nullptr); start.d[k] = std::strtof(argv[10 + k], nullptr); }
    const float radius = std::strtof(argv[13], nullptr);
    const int reps = argc > 14 ? std::atoi(argv[14]) : 7;
    std::vector<float> cloud(3 * n);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(cloud.data(), 4, 3 * n, f) != 3 * n) { std::printf("{\"error\": \"cannot read cloud\"}\n"); return 1; }
    std::fclose(f);
    Vector3f first;
    for (int k = 0; k < 3; ++k) first.d[k] = cloud[k];

    struct Run { std::vector<double> build, exp, cont, cost, route, total; size_t nodes = 0, cols = 0, slopes_touched = 0, steps = 0; bool found = false; unsigned levels = 0; };
    Run R[2];
    for (int mode = 0; mode < 2; ++mode) {             // 0 eager, 1 lazy
        TwoDmap map2D(gl, zl);
        map2D.setInterval(iv);
        map2D.setCloudFirst(first);
        for (int it = 0; it < reps + 2; ++it) {        // (two warm-up rounds: handle creation, first-build sizing)
            const auto t0 = Clock::now();
            if (!map2D.create2DMap(demand, cloud.data() + 3, n - 1, 12, mode == 1)) { std::printf("{\"error\": \"%s\"}\n", map2D.lastError().c_str()); return 1; }
            const auto t1 = Clock::now();
            RobotSphere robot(radius, start, goal);
            if (!map2D.computeCost(robot.getGoal(), robot, demand)) { std::printf("{\"error\": \"%s\"}\n", map2D.lastError().c_str()); return 1; }
            const auto t2 = Clock::now();
            AstarPlanar planner(robot.getPosition(), robot.getGoal());
            const bool found = planner.findRoute(map2D, robot, demand);
            const auto t3 = Clock::now();
            if (it < 2) continue;
            Run& r = R[mode];
            r.build.push_back(map2D.timing.build_ms); r.exp.push_back(map2D.timing.export_ms); r.cont.push_back(map2D.timing.containers_ms);
            r.cost.push_back(ms(t1, t2)); r.route.push_back(ms(t2, t3)); r.total.push_back(ms(t0, t3));
            r.nodes = map2D.exported().num_nodes; r.cols = map2D.exported().num_columns; r.found = found; r.steps = planner.global_path.size();
            r.levels = map2D.costStats().levels;
        }
    }
    std::printf("{\"points\": %zu, \"nodes\": %zu, \"columns\": %zu, \"route_found\": %s, \"route_steps\": %zu, \"flood_layers\": %u", n - 1, R[0].nodes,
                R[0].cols, R[0].found ? "true" : "false", R[0].steps, R[0].levels);
    const char* names[2] = {"eager", "lazy"};
    for (int mode = 0; mode < 2; ++mode) {
        const Run& r = R[mode];
        std::printf(", \"%s\": {\"build_sync_ms\": %.4f, \"export_ms\": %.4f, \"containers_ms\": %.4f, \"compute_cost_ms\": %.4f, \"find_route_ms\": %.4f, "
                    "\"total_ms\": %.4f, \"route_found\": %s, \"route_steps\": %zu}", names[mode], median(r.build), median(r.exp), median(r.cont),
                    median(r.cost), median(r.route), median(r.total), r.found ? "true" : "false", r.steps);
    }
    std::printf("}\n");
    return 0;
}
