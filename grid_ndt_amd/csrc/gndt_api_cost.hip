// gndt_api_cost.hip — cost-map flood (TwoDmap::computeCost, include/map2D.h:1285-1397) over the finished grid.
#include <atomic>

#include "gndt_handle.hpp"

using namespace gndt;
using namespace gndt_host;

namespace gndt_host {

void free_cost(gndt_handle* h) {
    auto& c = h->cost;
    void* ptrs[] = {c.h_bits, c.state, c.f[0], c.f[1], c.ctab_key, c.ctab_val, c.nbr, c.edges, c.d_cc};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c.h_cc) (void)hipHostFree(c.h_cc);
    c = gndt_handle::Cost{};
}

}  // namespace gndt_host

extern "C" {

constexpr int kCostBlocks = 128, kCostThreads = 64, kCostBatch = 32;

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
    if (n > c.node_cap) {
        for (uint32_t** a : {&c.h_bits, &c.state, &c.f[0], &c.f[1], &c.nbr}) { if (*a) (void)hipFree(*a); *a = nullptr; }
        if (c.edges) (void)hipFree(c.edges);
        c.edges = nullptr;
        c.node_cap = 0;
        c.tables_serial = 0;
        const uint64_t cap = std::max<uint64_t>(1024, n + n / 8);
        for (uint32_t** a : {&c.h_bits, &c.state, &c.f[0], &c.f[1]}) HIP_TRY(h, hipMalloc(a, cap * 4));
        // per row: 4 neighbour columns with their sizes (32 B), own column + collision verdict (8), the ring's step masks (16) and
        // two pairs of extremes for its rounds (16)
        HIP_TRY(h, hipMalloc(&c.nbr, cap * 72));
        HIP_TRY(h, hipMalloc(&c.edges, cap * 4 * sizeof(CostEdge)));
        c.node_cap = cap;
    }
    const uint32_t tsize = pow2_ceil(std::max<uint64_t>(1024, 2 * K));
    if (tsize > c.ctab_size) {
        if (c.ctab_key) (void)hipFree(c.ctab_key);
        if (c.ctab_val) (void)hipFree(c.ctab_val);
        c.ctab_key = nullptr; c.ctab_val = nullptr; c.ctab_size = 0;
        c.tables_serial = 0;
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
    V.nbr = nullptr; V.self = nullptr;
    V.slope_interval = h->P.slope_interval; V.demand_true = h->P.demand == GNDT_DEMAND_TRUE ? 1 : 0;
    // the goal's key through the same codec the build uses (transMortonXYZ, map2D.h:1293)
    const PointKey gk = point_key(goal_xyz[0], goal_xyz[1], goal_xyz[2], h->origin[0], h->origin[1], h->origin[2],
                                  h->P.grid_len, h->P.z_len);
    // What depends on the map and the robot only — the column index, every row's neighbour columns / own column / CostEdge records,
    // CollisionCheck's verdict for every slope — is kept from one flood to the next: the planner asks for a new goal on the same map
    // (receiver.cpp:160-176 floods once per goal message), and those passes are 50 of bridge_ground's 430 us.
    const float robot4[4] = {R.r, R.reach, R.rough, R.angle};
    // (Not on a handle that has recorded a hipGraph: a replay rewrites the map without the host's serial moving.)
    const bool tables_kept = K && !h->ever_captured && c.tables_serial == h->result_serial && std::memcmp(robot4, c.tables_robot, sizeof(robot4)) == 0;
    c.tables_serial = 0;
    uint32_t* self = c.nbr + 8 * c.node_cap;
    hipLaunchKernelGGL(k_cost_clear, dim3(grid_for(std::max<uint64_t>(n, tables_kept ? 0 : c.ctab_size))), dim3(256), 0, s, c.h_bits,
                       c.state, (uint32_t)n, c.ctab_key, tables_kept ? 0u : c.ctab_size, c.d_cc);
    if (K && !tables_kept) {
        hipLaunchKernelGGL(k_cost_columns, dim3(grid_for(n)), dim3(256), 0, s, h->out.sx, h->out.sy, h->part.row_ncol,
                           (uint32_t)n, c.ctab_key, c.ctab_val, c.ctab_size - 1, c.d_cc);
        // the per-map tables (gndt_cost.hpp): neighbour columns, own column, and CollisionCheck's verdict for every slope — with a
        // robot wider than a cell after ring_n rounds of "the extreme over my steps" over the whole map instead of a ring per slope
        uint32_t* step = c.nbr + 10 * c.node_cap;
        float* ext = reinterpret_cast<float*>(c.nbr + 14 * c.node_cap);            // hi[2], lo[2]: node_cap floats each
        float* hi[2] = {ext, ext + c.node_cap};
        float* lo[2] = {ext + 2 * c.node_cap, ext + 3 * c.node_cap};
        hipLaunchKernelGGL(k_cost_neighbours, dim3(grid_for(4 * n)), dim3(256), 0, s, V, R, (uint32_t)n, c.ring_n, c.nbr, self, step, hi[0], lo[0], c.edges);   // (probes: V.nbr, V.self are null)
        V.nbr = c.nbr;
        V.self = self;
        V.edges = c.edges;
        for (int d = 0; d < c.ring_n; ++d)
            hipLaunchKernelGGL(k_cost_ring_round, dim3(grid_for(4 * n)), dim3(256), 0, s, V, R, (uint32_t)n, step, hi[d & 1], hi[(d + 1) & 1],
                               lo[d & 1], lo[(d + 1) & 1], self, d == c.ring_n - 1 ? 1 : 0);
    }
    if (K) { V.nbr = c.nbr; V.self = self; V.edges = c.edges; }
    c.ring_store = c.ring_n > 0 ? 1 : 0;
    if (gk.ok && K)
        hipLaunchKernelGGL(k_cost_goal, dim3(1), dim3(64), 0, s, V, gk.sx, gk.sy, gk.sz, c.h_bits, c.f[0], c.d_cc);
    HIP_TRY(h, hipGetLastError());
    // A batch on the stream: the one-workgroup kernel (as many narrow layers as it meets, gndt_cost.hpp), then one-layer launches
    // (any width; no-ops once the flood has ended) — 8 while the layers are narrow, 32 when the last answer showed a wide one
    // (batches growing to 128 while the layers stay wide were measured on the 807-layer open site: no gain).  How many layers the
    // one-workgroup kernel walked is only known on the device (cc->wg_layers); the host asks after every batch whether the flood
    // has ended.  gndt_debug_set_option(GNDT_DEBUG_COST_ONE_WORKGROUP, 0): one-layer launches only.
    // (One workgroup takes ~5.6 us + 8 ns per slope for a layer, a one-layer launch 8-10 us whatever the width — site, terrain and a
    //  200 m open site with layers of thousands, profiles/r04_cost_map.json: the workgroup keeps the layers of up to kWgNarrow slopes.)
    constexpr uint32_t kWgNarrow = 320;
    const bool wg = tuning().cost_one_workgroup;
    // maps of up to kCostLdsRows rows: the one-workgroup kernel keeps h in LDS (144 KB of dynamic LDS have to be asked for once per
    // device: the attribute belongs to the function ON the current device, which check_ready has made the handle's)
    static std::atomic<int> lds_h_state[64];              // per device: 0 not asked yet, 1 granted, 2 refused
    bool lds_h_granted = false;
    if (h->device >= 0 && h->device < 64) {
        int st = lds_h_state[h->device].load(std::memory_order_acquire);
        if (st == 0) {
            st = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cost_flood_wg<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(kCostLdsRows * 4u)) == hipSuccess ? 1 : 2;
            if (st == 2) (void)hipGetLastError();
            lds_h_state[h->device].store(st, std::memory_order_release);
        }
        lds_h_granted = st == 1;
    }
    const bool lds_h = lds_h_granted && n <= kCostLdsRows;
    bool narrow = true, first_batch = true;
    uint32_t launched = 0;                                   // one-layer launches enqueued so far
    uint32_t blocks = kCostBlocks;                           // wavefronts of a one-layer launch (16 slopes at a time each): twice the last
                                                             //   layer seen, so that a wide layer is one pass
    for (;;) {
        if (wg && lds_h)
            hipLaunchKernelGGL(k_cost_flood_wg<true>, dim3(1), dim3(kWgThreads), (size_t)n * 4, s, V, R, c.h_bits, c.state, c.f[0], c.f[1], c.d_cc, kWgNarrow,
                               1u << 20, launched, (uint32_t)n, 1);
        else if (wg)
            hipLaunchKernelGGL(k_cost_flood_wg<false>, dim3(1), dim3(kWgThreads), 0, s, V, R, c.h_bits, c.state, c.f[0], c.f[1], c.d_cc, kWgNarrow,
                               1u << 20, launched, (uint32_t)n, 0);
        // (a flood the one-workgroup kernel walks to its end — bridge_ground, the site — pays 4.6 us for every one-layer launch behind
        //  it that finds nothing to do: two behind the first kernel, eight once a layer has been too wide for it)
        for (int b = 0, nb = !wg ? kCostBatch : first_batch ? 2 : narrow ? 8 : kCostBatch; b < nb; ++b, ++launched)
            hipLaunchKernelGGL(k_cost_level, dim3(blocks), dim3(kCostThreads), 0, s, V, R, c.h_bits, c.state, c.f[0], c.f[1], c.d_cc, launched);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(c.h_cc, c.d_cc, sizeof(CostCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        const uint32_t level = launched + c.h_cc->wg_layers;
        const uint32_t left = c.h_cc->frontier[level % 3u];
        if (left == 0u) break;
        first_batch = false;
        narrow = left <= kWgNarrow;
        blocks = (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(kCostBlocks, pow2_ceil((uint64_t)left / 8 + 1)));
        if (level > (1u << 24)) { h->err = "cost flood did not terminate"; return GNDT_ERR_HIP; }
    }
#if defined(GNDT_COST_STAMPS)
    std::fprintf(stderr, "[gndt cost stamps] layers in the one-workgroup kernel %llu; cycles per layer (wave 0): frontier read %.0f, loads %.0f, "
                 "minima %.0f, append %.0f, barrier + count %.0f\n", c.h_cc->phase[5], (double)c.h_cc->phase[0] / std::max<double>(1, c.h_cc->phase[5]),
                 (double)c.h_cc->phase[1] / std::max<double>(1, c.h_cc->phase[5]), (double)c.h_cc->phase[2] / std::max<double>(1, c.h_cc->phase[5]),
                 (double)c.h_cc->phase[3] / std::max<double>(1, c.h_cc->phase[5]), (double)c.h_cc->phase[4] / std::max<double>(1, c.h_cc->phase[5]));
#endif
    if (c.h_cc->range_error) {
        h->err = "cost map: column indices beyond 32767 (mortonToXY decodes no further, Stopwatch.h:171-189)";
        return GNDT_ERR_KEY_RANGE;
    }
    if (n) {        // what no relaxation reached keeps the FLT_MAX it was created with (gndt_cost.hpp: kUnreachedBits)
        hipLaunchKernelGGL(k_cost_finish, dim3(grid_for(n)), dim3(256), 0, s, c.h_bits, c.state, V.self, (uint32_t)n);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    c.serial = h->result_serial;
    if (K) { c.tables_serial = h->result_serial; std::memcpy(c.tables_robot, robot4, sizeof(robot4)); }     // (range_error would have returned above)
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
