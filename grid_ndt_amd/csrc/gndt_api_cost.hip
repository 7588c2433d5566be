// gndt_api_cost.hip — cost-map flood (TwoDmap::computeCost, include/map2D.h:1285-1397) over the finished grid.
#include "gndt_handle.hpp"

using namespace gndt;
using namespace gndt_host;

namespace gndt_host {

void free_cost(gndt_handle* h) {
    auto& c = h->cost;
    void* ptrs[] = {c.h_bits, c.state, c.f[0], c.f[1], c.ctab_key, c.ctab_val, c.ring, c.nbr, c.d_cc};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c.h_cc) (void)hipHostFree(c.h_cc);
    c = gndt_handle::Cost{};
}


static int env_int(const char* name, int otherwise) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : otherwise;
}

}  // namespace gndt_host

extern "C" {

constexpr int kCostBlocks = 128, kCostThreads = 64, kCostBatch = 32;
constexpr int kCostWaves = 512;      // wavefronts of a layer launch whose slopes are checked by a whole wavefront each (rings in global scratch)

int gndt_compute_cost(gndt_handle* h, const float goal_xyz[3], const gndt_robot* robot, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!goal_xyz) { h->err = "null goal"; return GNDT_ERR_INVALID; }
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to flood (computeCost runs after create2DMap, receiver.cpp:160, 171)"; return GNDT_ERR_INVALID; }
    rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    hipStream_t s = stream_of(h, hip_stream);
    auto& c = h->cost;
    c.serial = 0;
    const uint64_t n = h->res_nodes, K = h->res_columns;
    if (!c.d_cc) {
        HIP_TRY(h, hipMalloc(&c.d_cc, sizeof(CostCounters)));
        HIP_TRY(h, hipHostMalloc(&c.h_cc, sizeof(CostCounters)));
    }
    // rings of up to lds_cap slopes stay in LDS; larger ones go to global scratch, ring_cap slopes per checker at first
    // (GNDT_COST_RING_LDS / GNDT_COST_RING_FIRST: smaller numbers for the tests of exactly these steps)
    const int lds_cap = std::max(1, std::min(kTeamRingCap, env_int("GNDT_COST_RING_LDS", kTeamRingCap)));
    if (c.ring_cap == 0) c.ring_cap = (int)pow2_ceil((uint64_t)std::max(16, std::min(kRingCapMax, env_int("GNDT_COST_RING_FIRST", 8 * kTeamRingCap))));
    if (n > c.node_cap) {
        for (uint32_t** a : {&c.h_bits, &c.state, &c.f[0], &c.f[1], &c.nbr}) { if (*a) (void)hipFree(*a); *a = nullptr; }
        c.node_cap = 0;
        const uint64_t cap = std::max<uint64_t>(1024, n + n / 8);
        for (uint32_t** a : {&c.h_bits, &c.state, &c.f[0], &c.f[1]}) HIP_TRY(h, hipMalloc(a, cap * 4));
        HIP_TRY(h, hipMalloc(&c.nbr, cap * 40));      // (4 neighbour columns with their sizes + own column + verdict on the slope above, per row)
        c.node_cap = cap;
    }
    const uint32_t tsize = pow2_ceil(std::max<uint64_t>(1024, 2 * K));
    if (tsize > c.ctab_size) {
        if (c.ctab_key) (void)hipFree(c.ctab_key);
        if (c.ctab_val) (void)hipFree(c.ctab_val);
        c.ctab_key = nullptr; c.ctab_val = nullptr; c.ctab_size = 0;
        HIP_TRY(h, hipMalloc(&c.ctab_key, (size_t)tsize * 8));
        HIP_TRY(h, hipMalloc(&c.ctab_val, (size_t)tsize * 4));
        c.ctab_size = tsize;
    }
    Robot R{0.25f, 0.15f, 100.f, 30.f};   // receiver.cpp:33, robot.h:38-46
    if (robot) R = Robot{robot->radius, robot->reachable_height, robot->max_rough, robot->max_angle_deg};
    c.ring_n = cost_ring_depth(R.r, h->P.grid_len);
    CostView V;
    V.sx = h->out.sx; V.sy = h->out.sy; V.sz = h->out.sz;
    V.mean = h->out.mean; V.normal = h->out.normal; V.rough = h->out.rough; V.flags = h->out.flags;
    V.row_ncol = h->part.row_ncol;
    V.ctab_key = c.ctab_key; V.ctab_val = c.ctab_val; V.ctab_mask = c.ctab_size - 1;
    V.nbr = nullptr;
    V.slope_interval = h->P.slope_interval; V.demand_true = h->P.demand == GNDT_DEMAND_TRUE ? 1 : 0;
    // the goal's key through the same codec the build uses (transMortonXYZ, map2D.h:1293)
    const PointKey gk = point_key(goal_xyz[0], goal_xyz[1], goal_xyz[2], h->origin[0], h->origin[1], h->origin[2],
                                  h->P.grid_len, h->P.z_len);
  for (;;) {                  // (again with a larger ring scratch if a collision ring did not fit)
    V.nbr = nullptr; V.self = nullptr;
    hipLaunchKernelGGL(k_cost_clear, dim3(grid_for(std::max<uint64_t>(n, c.ctab_size))), dim3(256), 0, s, c.h_bits,
                       c.state, (uint32_t)n, c.ctab_key, c.ctab_size, c.d_cc);
    if (K)
        hipLaunchKernelGGL(k_cost_columns, dim3(grid_for(n)), dim3(256), 0, s, h->out.sx, h->out.sy, h->part.row_ncol,
                           (uint32_t)n, c.ctab_key, c.ctab_val, c.ctab_size - 1, c.d_cc);
    if (K) {
        hipLaunchKernelGGL(k_cost_neighbours, dim3(grid_for(4 * n)), dim3(256), 0, s, V, R, (uint32_t)n, c.nbr, c.nbr + 8 * c.node_cap);   // (probes: V.nbr, V.self are null)
        V.nbr = c.nbr;
        V.self = c.nbr + 8 * c.node_cap;
    }
    if (gk.ok && K)
        hipLaunchKernelGGL(k_cost_goal, dim3(1), dim3(64), 0, s, V, gk.sx, gk.sy, gk.sz, c.h_bits, c.f[0], c.d_cc);
    HIP_TRY(h, hipGetLastError());
    // Who checks a slope for collisions (gndt_cost.hpp): without a ring one lane; with a ring a team of 16 and the ring in LDS, or —
    // once a ring of that depth has not fitted on this handle — a whole wavefront per slope with ring and set in global scratch,
    // which is only allocated then.  GNDT_COST_RING_IN_LDS=0 sends every ring the second way (tests).
    const bool lds_off = env_int("GNDT_COST_RING_IN_LDS", 1) == 0;
    const int mode = c.ring_n == 0 ? 0 : (c.ring_n < c.team_ring_limit && !lds_off) ? 1 : 2;
    c.ring_store = mode;
    if (mode == 2 && c.ring_alloc < c.ring_cap) {
        if (c.ring) (void)hipFree(c.ring);
        c.ring = nullptr; c.ring_alloc = 0;
        const size_t words = (size_t)kCostWaves * 3u * (size_t)c.ring_cap;                       // (ring + a set twice as large, per wavefront)
        HIP_TRY(h, hipMalloc(&c.ring, words * sizeof(uint32_t)));
        HIP_TRY(h, hipMemsetAsync(c.ring, 0xFF, words * sizeof(uint32_t), s));                   // (sets are empty at rest)
        c.ring_alloc = c.ring_cap;
    }
    // A batch on the stream: without collision rings the one-workgroup kernel (as many narrow layers as it meets, gndt_cost.hpp), then
    // one-layer launches (any width; no-ops once the flood has ended) — 8 while the layers are narrow, 32 when the last answer showed a
    // wide one.  How many layers the one-workgroup kernel walked is only known on the device (cc->wg_layers); the host asks after
    // every batch whether the flood has ended.  GNDT_COST_WG=0: one-layer launches only.
    const bool wg = mode == 0 && env_int("GNDT_COST_WG", 1) != 0;
    bool narrow = true;
    uint32_t launched = 0;                                   // one-layer launches enqueued so far
    for (;;) {
        if (wg)
            hipLaunchKernelGGL(k_cost_flood_wg, dim3(1), dim3(kWgThreads), 0, s, V, R, c.h_bits, c.state, c.f[0], c.f[1], c.d_cc, kWgFrontier, 1u << 20,
                               launched);
        for (int b = 0, nb = wg && narrow ? 8 : kCostBatch; b < nb; ++b, ++launched) {
            if (mode == 1)
                hipLaunchKernelGGL(k_cost_level<16>, dim3(kCostBlocks * 4), dim3(kCostThreads), 0, s, V, R, c.ring_n, c.h_bits,
                                   c.state, c.f[0], c.f[1], (uint32_t*)nullptr, (uint32_t)lds_cap, c.d_cc, launched);
            else if (mode == 2)
                hipLaunchKernelGGL(k_cost_level<64>, dim3(kCostWaves), dim3(kCostThreads), 0, s, V, R, c.ring_n, c.h_bits,
                                   c.state, c.f[0], c.f[1], c.ring, (uint32_t)c.ring_cap, c.d_cc, launched);
            else
                hipLaunchKernelGGL(k_cost_level<4>, dim3(kCostBlocks), dim3(kCostThreads), 0, s, V, R, 0, c.h_bits,
                                   c.state, c.f[0], c.f[1], (uint32_t*)nullptr, 1u, c.d_cc, launched);
        }
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(c.h_cc, c.d_cc, sizeof(CostCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        const uint32_t level = launched + c.h_cc->wg_layers;
        const uint32_t left = c.h_cc->frontier[level % 3u];
        if (left == 0u) break;
        narrow = left <= kWgFrontier;
        if (level > (1u << 24)) { h->err = "cost flood did not terminate"; return GNDT_ERR_HIP; }
    }
    if (c.h_cc->range_error) {
        h->err = "cost map: column indices beyond 32767 (mortonToXY decodes no further, Stopwatch.h:171-189)";
        return GNDT_ERR_KEY_RANGE;
    }
    if (c.h_cc->ring_overflow) {
        // CollisionCheck's ring is a std::list in the reference (map2D.h:351-411): any size.  Here it is storage of a fixed size per
        // checker, made four times larger and the flood run again when a ring did not fit.
        if (mode == 1) { c.team_ring_limit = c.ring_n; continue; }     // (this handle's maps hold rings of this depth that do not fit LDS)
        if (c.ring_cap >= kRingCapMax) {
            h->err = "cost map: a collision ring holds more than " + std::to_string(kRingCapMax) + " slopes (robot radius too large for this grid)";
            return GNDT_ERR_CAPACITY;
        }
        c.ring_cap = std::min(c.ring_cap * 4, kRingCapMax);
        continue;
    }
    break;
  }
    if (n) {        // what no relaxation reached keeps the FLT_MAX it was created with (gndt_cost.hpp: kUnreachedBits)
        hipLaunchKernelGGL(k_cost_finish, dim3(grid_for(n)), dim3(256), 0, s, c.h_bits, (uint32_t)n);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    c.serial = h->result_serial;
    return GNDT_OK;
}

static int cost_ready(gndt_handle* h, gndt_cost_stats* st) {
    if (!h) return GNDT_ERR_INVALID;
    if (!h->results_valid || h->cost.serial == 0 || h->cost.serial != h->result_serial) {
        h->err = "no cost map for the current grid (call gndt_compute_cost after the build)";
        return GNDT_ERR_INVALID;
    }
    if (st) {
        const CostCounters* cc = h->cost.h_cc;
        st->goal_status = cc->goal_status; st->ring = (uint32_t)h->cost.ring_n; st->levels = cc->levels; st->ring_store = (uint32_t)h->cost.ring_store;
        st->traversable = cc->traversable; st->closed = cc->closed; st->check_pushes = cc->check_pushes;
    }
    return GNDT_OK;
}

int gndt_cost_export_device(gndt_handle* h, const float** h_dev, const uint32_t** state_dev, gndt_cost_stats* stats) {
    int rc = cost_ready(h, stats);
    if (rc) return rc;
    if (h_dev) *h_dev = reinterpret_cast<const float*>(h->cost.h_bits);
    if (state_dev) *state_dev = h->cost.state;
    return GNDT_OK;
}

int gndt_cost_export(gndt_handle* h, float* h_out, uint32_t* state_out, gndt_cost_stats* stats) {
    int rc = cost_ready(h, stats);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const uint64_t n = h->res_nodes;
    if (h_out && n) HIP_TRY(h, hipMemcpy(h_out, h->cost.h_bits, n * 4, hipMemcpyDeviceToHost));
    if (state_out && n) HIP_TRY(h, hipMemcpy(state_out, h->cost.state, n * 4, hipMemcpyDeviceToHost));
    return GNDT_OK;
}

}  // extern "C"
