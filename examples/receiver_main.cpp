// receiver_main.cpp — a ROS-free rendition of the reference node's callback (src/receiver.cpp:137-176) on top of
// libgndt's C ABI and the reference-shaped containers of include/gndt_compat.hpp:
//
//     .pcd file            -> gndt_pcd_read                                   (publisher.cpp:15-36 loadCloud)
//     raw records          -> TwoDmap::create2DMapFromRaw                     (receiver.cpp:140-160: NaN strip, origin =
//                                                                              first point, uniformDivision loop, create2DMap)
//     goal                 -> TwoDmap::computeCost                            (receiver.cpp:171, map2D.h:1285-1397; on the GPU)
//     start, goal          -> AstarPlanar::findRoute                          (receiver.cpp:173-175, GlobalPlan.h:49-166)
//
//   receiver_main <cloud.pcd> <gridLen> <zLen> <slope_interval> <demand> <gx> <gy> <gz> <sx> <sy> <sz> <robot radius>
// prints one line per stage (the reference prints its own timings the same way) and the route as "key z h" lines.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "gndt_compat.hpp"

using namespace gndt_compat;

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 13) {
        std::printf("usage: %s cloud.pcd gridLen zLen slope_interval demand gx gy gz sx sy sz radius\n", argv[0]);
        return 2;
    }
    const float gridLen = std::strtof(argv[2], nullptr), zLen = std::strtof(argv[3], nullptr), interval = std::strtof(argv[4], nullptr);
    const std::string demand = argv[5];
    Vector3f goal, start;
    for (int k = 0; k < 3; ++k) { goal.d[k] = std::strtof(argv[6 + k], nullptr); start.d[k] = std::strtof(argv[9 + k], nullptr); }
    const float radius = std::strtof(argv[12], nullptr);

    gndt_pcd pcd;
    char err[256] = {0};
    double t0 = now_s();
    if (gndt_pcd_read(argv[1], &pcd, err) != GNDT_OK) { std::printf("ERROR: %s\n", err); return 1; }
    std::printf("Loaded %llu points (%s, %u-byte records) in %.3f s\n", (unsigned long long)pcd.num_points,
                pcd.data_kind == 0 ? "ascii" : pcd.data_kind == 1 ? "binary" : "binary_compressed", pcd.layout.point_step, now_s() - t0);

    TwoDmap map2D(gridLen, zLen);
    map2D.setInterval(interval);
    t0 = now_s();
    if (!map2D.create2DMapFromRaw(demand, pcd.data, pcd.num_points, pcd.layout)) { std::printf("wrong: %s\n", map2D.lastError().c_str()); return 1; }
    size_t slopes = 0;
    for (auto& kv : map2D.map_cell) slopes += kv.second->map_slope.size();
    std::printf("division + create2DMap: %.3f s  nodes %zu columns %zu slopes %zu\n", now_s() - t0, map2D.map_xy.size(),
                map2D.map_cell.size(), slopes);
    gndt_pcd_free(&pcd);

    RobotSphere robot(radius, start, goal);
    t0 = now_s();
    if (!map2D.computeCost(robot.getGoal(), robot, demand)) { std::printf("wrong: %s\n", map2D.lastError().c_str()); return 1; }
    const gndt_cost_stats& cs = map2D.costStats();
    if (cs.goal_status == 2) std::printf("Goal position wrong: cant find goal slope.\n");
    std::printf("traversability time: %.3f s  traversability slopes %llu  check slopes %llu  layers %u\n", now_s() - t0,
                (unsigned long long)cs.traversable, (unsigned long long)cs.check_pushes, cs.levels);

    AstarPlanar planner(robot.getPosition(), robot.getGoal());
    t0 = now_s();
    const bool found = planner.findRoute(map2D, robot, demand);
    std::printf("%s  A*: %.3f s  steps %zu\n", found ? "found the route to goal" : "not find the road", now_s() - t0,
                planner.global_path.size());
    for (const Slope* s : planner.global_path) std::printf("route %s %d %.9g\n", s->morton_xy.c_str(), s->morton_z, (double)s->h);
    return found ? 0 : 3;
}
