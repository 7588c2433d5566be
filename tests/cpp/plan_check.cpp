// Test program for the consumers of the grid in include/gndt_compat.hpp (built and run by tests/test_compat_cpp.py):
//   plan_check <cloud.f32> <n> <grid_len> <z_len> <interval> <demand> <gx> <gy> <gz> <sx> <sy> <sz> <radius> [gpu]
// 1. CPU: oracle grid -> materialise; the oracle's computeCost + AstarPlanar (liboracle.so: test infrastructure)
//    give h and the route; h is written into the containers and gndt_compat::AstarPlanar must return the same route.
// 2. "gpu": gndt_compat::TwoDmap::create2DMap + computeCost run on the GPU (libgndt, C ABI); every Slope::h must
//    equal the oracle's flood of the SAME exported grid bit for bit, and the planner's route must be the oracle's.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gndt_compat.hpp"

extern "C" {
void* oracle_build(const float* xyz, size_t n, size_t stride_floats, const float origin[3], float grid_len, float z_len,
                   float slope_interval, int demand, int min_points, int mode, int threads);
size_t oracle_num_nodes(void* h);
size_t oracle_num_columns(void* h);
void oracle_export(void* h, int32_t* sx, int32_t* sy, int32_t* sz, uint32_t* count, uint64_t* first_idx, float* mean,
                   float* cov, float* evals, float* rough, float* normal, uint32_t* flags, double* mean64, double* cov64,
                   double* rough64, double* normal64, double* evals64, char* morton);
void oracle_free(void* h);
int oracle_compute_cost(size_t n, const int32_t* sx, const int32_t* sy, const int32_t* sz, const uint32_t* count,
                        const float* mean, const float* normal, const float* rough, const uint32_t* flags, const float origin[3],
                        float grid_len, float z_len, float slope_interval, int demand_true, const float goal[3],
                        const float robot4[4], int mode, float* h_out, uint8_t* state_out, int64_t stats[4], double margins[2],
                        const float* start, int32_t* path_rows, int64_t path_cap, int64_t* path_len);
}

using namespace gndt_compat;

#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

struct RefPlan {
    std::vector<float> h;
    std::vector<uint8_t> state;
    std::vector<int32_t> path;
    int64_t stats[4];
    int rc;
};

static RefPlan reference_plan(const gndt_cells& c, const float* origin, float gl, float zl, float iv, bool demand_true,
                              const float goal[3], const float start[3], float radius) {
    RefPlan r;
    const size_t n = c.num_nodes;
    r.h.resize(n); r.state.resize(n); r.path.resize(n ? n : 1);
    const float robot4[4] = {radius, 0.15f, 100.f, 30.f};
    double margins[2];
    int64_t len = 0;
    r.rc = oracle_compute_cost(n, c.sx, c.sy, c.sz, c.count, c.mean, c.normal, c.rough, c.flags, origin, gl, zl, iv,
                               demand_true ? 1 : 0, goal, robot4, 1, r.h.data(), r.state.data(), r.stats, margins, start,
                               r.path.data(), (int64_t)n, &len);
    r.path.resize((size_t)len);
    return r;
}

// PLAN_PRINT=1: the planner's route as "key/z key/z ..." (start -> goal) and Slope::h at the start, so that a test can
// hold it against a route derived by hand (tests/test_planner_hand_routes.py)
static void print_route(const std::list<Slope*>& got) {
    if (!std::getenv("PLAN_PRINT")) return;
    std::printf("route:");
    for (const Slope* s : got) std::printf(" %s/%d", s->morton_xy.c_str(), s->morton_z);
    std::printf("\nh_start: %.9g\n", got.empty() ? -1.0 : (double)got.front()->h);
}

static int same_route(const std::list<Slope*>& got, const RefPlan& ref, const gndt_cells& c, const char* what) {
    print_route(got);
    CHECK(got.size() == ref.path.size(), "%s: route length %zu != oracle %zu", what, got.size(), ref.path.size());
    size_t k = 0;
    for (const Slope* s : got) {
        const int32_t row = ref.path[k++];
        CHECK(s->morton_xy == column_key(c.sx[row], c.sy[row]) && s->morton_z == c.sz[row], "%s: step %zu is %s/%d", what, k - 1,
              s->morton_xy.c_str(), s->morton_z);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 14) { std::printf("usage\n"); return 2; }
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    const float gl = std::strtof(argv[3], nullptr), zl = std::strtof(argv[4], nullptr), iv = std::strtof(argv[5], nullptr);
    const std::string demand = argv[6];
    float goal[3], start[3];
    for (int k = 0; k < 3; ++k) { goal[k] = std::strtof(argv[7 + k], nullptr); start[k] = std::strtof(argv[10 + k], nullptr); }
    const float radius = std::strtof(argv[13], nullptr);
    const bool gpu = argc > 14 && std::string(argv[14]) == "gpu";
    std::vector<float> cloud(3 * n);
    FILE* f = std::fopen(argv[1], "rb");
    CHECK(f && std::fread(cloud.data(), 4, 3 * n, f) == 3 * n, "cannot read cloud");
    std::fclose(f);
    const Vector3f vgoal{{goal[0], goal[1], goal[2]}}, vstart{{start[0], start[1], start[2]}};

    if (!gpu) {
        void* oh = oracle_build(cloud.data() + 3, n - 1, 3, cloud.data(), gl, zl, iv, demand == "true" ? 1 : 0, 3, 1, 1);
        const size_t C = oracle_num_nodes(oh);
        CellsHost ex;
        ex.resize(C);
        std::vector<uint64_t> first64(C);
        std::vector<char> morton(16 * C);
        oracle_export(oh, ex.sx.data(), ex.sy.data(), ex.sz.data(), ex.count.data(), first64.data(), ex.mean.data(), ex.cov.data(),
                      nullptr, ex.rough.data(), ex.normal.data(), ex.flags.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                      morton.data());
        ex.view.num_columns = oracle_num_columns(oh);
        oracle_free(oh);
        TwoDmap A(gl, zl);
        A.setInterval(iv);
        A.setCloudFirst(Vector3f{{cloud[0], cloud[1], cloud[2]}});
        materialise(ex.view, A);
        const RefPlan ref = reference_plan(ex.view, cloud.data(), gl, zl, iv, demand == "true", goal, start, radius);
        CHECK(ref.rc == 0 && !ref.path.empty(), "oracle found no route (rc %d)", ref.rc);
        apply_cost_into(ex.view, ref.h.data(), A);
        RobotSphere robot(radius, vstart, vgoal);
        AstarPlanar planner(robot.getPosition(), robot.getGoal());
        CHECK(planner.findRoute(A, robot, demand), "compat planner found no route");
        if (same_route(planner.global_path, ref, ex.view, "cpu")) return 1;
        std::printf("compat A* == oracle OK steps=%zu h_start=%g\n", ref.path.size(), (double)ref.h[ref.path[0]]);
        return 0;
    }

    TwoDmap B(gl, zl);
    B.setInterval(iv);
    B.setCloudFirst(Vector3f{{cloud[0], cloud[1], cloud[2]}});
    CHECK(B.create2DMap(demand, cloud.data() + 3, n - 1, 12), "create2DMap failed: %s", B.lastError().c_str());
    RobotSphere robot(radius, vstart, vgoal);
    CHECK(B.computeCost(robot.getGoal(), robot, demand), "computeCost failed: %s", B.lastError().c_str());
    const gndt_cells& ex = B.exported();
    const RefPlan ref = reference_plan(ex, cloud.data(), gl, zl, iv, demand == "true", goal, start, radius);
    CHECK(ref.rc == 0 && B.costStats().goal_status == 0, "goal status %d / %d", ref.rc, B.costStats().goal_status);
    CHECK((int64_t)B.costStats().traversable == ref.stats[0] && (int64_t)B.costStats().closed == ref.stats[1] &&
              (int64_t)B.costStats().check_pushes == ref.stats[2], "flood statistics differ");
    size_t slopes = 0;
    for (uint64_t i = 0; i < ex.num_nodes; ++i) {
        if (!(ex.flags[i] & GNDT_FLAG_SLOPE)) continue;
        const Slope* s = B.map_cell.at(column_key(ex.sx[i], ex.sy[i]))->map_slope.at((int)ex.sz[i]);
        CHECK(std::memcmp(&s->h, &ref.h[i], 4) == 0, "row %llu: Slope::h %g != oracle %g", (unsigned long long)i, (double)s->h, (double)ref.h[i]);
        ++slopes;
    }
    AstarPlanar planner(robot.getPosition(), robot.getGoal());
    CHECK(planner.findRoute(B, robot, demand) == !ref.path.empty(), "route found / not found differs");
    if (same_route(planner.global_path, ref, ex, "gpu")) return 1;
    // ---- the same through the LAZY mode: no containers, the consumers served from the exported rows ----
    TwoDmap L(gl, zl);
    L.setInterval(iv);
    L.setCloudFirst(Vector3f{{cloud[0], cloud[1], cloud[2]}});
    CHECK(L.create2DMap(demand, cloud.data() + 3, n - 1, 12, true), "lazy create2DMap failed: %s", L.lastError().c_str());
    CHECK(L.isLazy() && L.map_cell.empty() && L.map_xy.empty() && L.morton_list.empty(), "lazy mode filled the containers");
    RobotSphere robot2(radius, vstart, vgoal);
    CHECK(L.computeCost(robot2.getGoal(), robot2, demand), "lazy computeCost failed: %s", L.lastError().c_str());
    AstarPlanar planner2(robot2.getPosition(), robot2.getGoal());
    CHECK(planner2.findRoute(L, robot2, demand) == !ref.path.empty(), "lazy: route found / not found differs");
    if (same_route(planner2.global_path, ref, ex, "lazy")) return 1;
    {
        size_t k = 0;
        for (const Slope* s : planner2.global_path) {
            const int32_t row = ref.path[k++];
            CHECK(std::memcmp(&s->h, &ref.h[row], 4) == 0, "lazy: h of step %zu differs", k - 1);
        }
        // field for field what the rows L was built from hold (two builds of one cloud agree in keys and order, and in the fp32
        // values up to the last bit: the order of the fp64 LDS additions is not fixed — so L's slopes are held to L's own rows)
        const gndt_cells& lx = L.exported();
        CHECK(lx.num_nodes == ex.num_nodes, "lazy build has %llu rows, eager %llu", (unsigned long long)lx.num_nodes, (unsigned long long)ex.num_nodes);
        k = 0;
        for (const Slope* s : planner2.global_path) {
            const int32_t row = ref.path[k++];
            CHECK(lx.sx[row] == ex.sx[row] && lx.sy[row] == ex.sy[row] && lx.sz[row] == ex.sz[row], "row %d: keys differ between two builds", row);
            CHECK(s->rough == lx.rough[row] && s->mean(2) == lx.mean[3 * row + 2] && s->normal(0) == lx.normal[3 * row] &&
                      s->down == ((lx.flags[row] & GNDT_FLAG_DOWN) != 0), "lazy slope of row %d differs from its row", row);
        }
    }
    std::printf("libgndt computeCost + A* == oracle OK slopes=%zu steps=%zu (eager and lazy)\n", slopes, ref.path.size());
    return 0;
}
