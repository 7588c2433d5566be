// gndt_api.hip — the C ABI of include/gndt.h over the kernels of gndt_kernels.hpp.
// Host code here only owns memory, orders launches on a HIP stream and maps errors to status codes.
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// GNDT_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "gndt.h"
#include "gndt_kernels.hpp"
#include "gndt_partition.hpp"
#include "gndt_bucket.hpp"
#include "gndt_table.hpp"
#include "gndt_cost.hpp"
#include "gndt_pack.hpp"

using namespace gndt;

namespace {
thread_local std::string g_create_error;
}

struct gndt_handle {
    gndt_params P{};
    float origin[3] = {0, 0, 0};
    bool origin_set = false;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;

    // node table
    uint32_t cap = 0;  // slots, power of two
    uint64_t* keys = nullptr;
    NodeAcc* acc = nullptr;
    uint64_t* col_keys = nullptr;
    uint32_t* col_first = nullptr;
    SlotAux* aux = nullptr;
    // node list + ordering buffers (cap entries each)
    uint32_t* node_slot = nullptr;
    uint32_t* col_slot_of_node = nullptr;
    uint32_t* col_cnt = nullptr;
    uint32_t* col_head = nullptr;
    uint32_t* node_next = nullptr;
    // incremental updates: slot -> node index, touch marks, the lists of touched nodes / columns
    uint32_t *index_of_slot = nullptr, *touch_epoch = nullptr, *col_epoch = nullptr, *touched = nullptr, *touched_cols = nullptr;
    bool incr_ok = false;       // the persistent staging rows / order of the table path describe the current map

    Counters* d_cnt = nullptr;
    Counters* h_cnt = nullptr;  // pinned

    // results
    uint64_t out_cap = 0;
    OutView out{};
    uint64_t res_nodes = 0, res_columns = 0, res_slopes = 0;
    bool results_valid = false;

    // stats export buffers
    uint64_t st_cap = 0;
    uint64_t* st_key = nullptr;
    double* st_sums = nullptr;
    uint32_t* st_count = nullptr;
    uint32_t* st_first = nullptr;

    // staging for host input
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // gndt_build_cloud: packed xyz of the unpacked, NaN-stripped cloud; pinned word for the valid count
    float* packed = nullptr;
    uint64_t packed_cap = 0;
    uint32_t* d_nvalid = nullptr;
    uint32_t* h_nvalid = nullptr;

    uint64_t stream_pos = 0;    // points accumulated since the last reset (host mirror of the device-side first_idx base)
    uint64_t nodes_bound = 0;   // host-side upper bound of the nodes in the table (no sync needed to size buffers)
    bool table_dirty = false;   // table holds nodes

    // strategy PARTITION buffers (gndt_partition.hpp)
    struct Part {
        uint64_t rec_cap = 0;      float4* recs = nullptr;
        // two-level partition: level-1 regions, cursors of both levels, record ranges of the fine buckets
        uint64_t rec1_cap = 0;     float4* recs1 = nullptr;
        uint64_t cur_cap = 0;      uint32_t *cursors = nullptr, *range_lo = nullptr, *range_hi = nullptr, *range_cap = nullptr;
        int two_level_failures = 0;   // builds whose regions overflowed although sized from the sample
        bool two_level_ok = true;  // cleared when the regions a cloud needs are too large: exact path from then on
        double fill1_ratio = 0.0;   // fullest level-1 region / mean seen on this handle (0 = unknown)
        uint64_t hist_cap = 0;     uint32_t* hist = nullptr;
        uint32_t bucket_cap = 0;   uint32_t* totals = nullptr; uint32_t* bucket_base = nullptr;
        uint64_t stage_cap = 0;    StageRow* stage = nullptr;
        uint32_t *ord_cf = nullptr, *ord_idx = nullptr, *inv = nullptr, *row_ncol = nullptr;
        // column order (gndt_partition.hpp ColumnOrder): per bitmap word, and per point index for ncol_at
        uint64_t words_init = 0;   // bitmap / word_weight words the table path's column order has initialised
        uint64_t word_cap = 0;     uint32_t *bitmap = nullptr, *word_weight = nullptr, *word_base = nullptr, *bsum_words = nullptr,
                                            *ncol_at = nullptr;
        PartCounters* d_pc = nullptr;
        PartCounters* h_pc = nullptr;   // pinned
        unsigned long long* dbg = nullptr;  uint32_t dbg_buckets = 0;   // diagnostic phase stamps (GNDT_STAMPS=1)
        uint32_t last_buckets = 0;
        uint64_t nodes_learned = 0;   // node count of the last successful PARTITION build (+20 %)
        int good_slots = 0; uint64_t good_est = 0, good_n = 0;   // table size / estimate that worked last time
    } part;
    // cost-map flood over the finished grid (gndt_cost.hpp)
    struct Cost {
        uint64_t node_cap = 0;     uint32_t *h_bits = nullptr, *pushed = nullptr, *state = nullptr, *f[2] = {nullptr, nullptr};
        uint32_t ctab_size = 0;    uint64_t* ctab_key = nullptr; uint32_t* ctab_val = nullptr;
        uint32_t* ring = nullptr;
        uint32_t* nbr = nullptr;       // [4 * node_cap] neighbour columns of every row
        CostCounters* d_cc = nullptr;
        CostCounters* h_cc = nullptr;   // pinned
        uint64_t serial = 0;            // result_serial the flood was computed for (0 = none)
        int ring_n = 0;
    } cost;
    uint64_t result_serial = 0;         // bumped whenever a build / finalize produces new result rows
    int last_strategy = GNDT_STRATEGY_ATOMIC;
    bool map_in_table = true;   // false after a PARTITION build: the HBM node table does not hold the map

    // optional phase timing (bench / profiling): events recorded on the launch stream
    int prof = 0;               // 0 off, 1 every phase, 2 only the dominant phase of the strategy in use
    // one event set per build in a ring, so that back-to-back (un-synchronised) builds can all be timed
    static constexpr int kEvSets = 32;
    hipEvent_t ev[kEvSets][GNDT_NUM_PHASES + 1] = {};
    bool ev_recorded[kEvSets][GNDT_NUM_PHASES + 1] = {};
    int ev_set = 0;

    // A PARTITION build is launched without waiting for it; its overflow flags are looked at (and the build
    // re-run with more room if they are set) by the next call that needs the result.
    struct Pending {
        bool active = false;
        const void* xyz = nullptr; size_t n = 0, stride = 0;
        hipStream_t s = nullptr;
        int attempt = 0, bslots = 0;
        uint64_t nodes_est = 0, stage_want = 0;
        bool two_level = false;         // this attempt used the two-level partition
        double mean1 = 0.0;             // its mean level-1 region fill (to turn the fullest region into a ratio)
        bool stats_only = false;        // gndt_shard_stats_device: statistics out, no labels / ordering / rows
        uint32_t first_base = 0;        // global index of xyz[0] (shards of a global cloud)
    } pending;

    std::string err;
};

namespace {

#define HIP_TRY(h, expr)                                                                                 \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                               \
            return GNDT_ERR_HIP;                                                                         \
        }                                                                                                \
    } while (0)

inline void mark(gndt_handle* h, int i, hipStream_t s) {
    if (!h->prof || !h->ev[h->ev_set][i]) return;
    if (h->prof == 2) {   // k_bucket_build2 sits between marks 4 and 5, k_accumulate between 1 and 2
        const int lo = h->last_strategy != GNDT_STRATEGY_ATOMIC ? 4 : 1;
        if (i != lo && i != lo + 1) return;
    }
    (void)hipEventRecord(h->ev[h->ev_set][i], s);
    h->ev_recorded[h->ev_set][i] = true;
}

// a new build / update starts: next event set of the ring
inline void next_event_set(gndt_handle* h) {
    if (!h->prof) return;
    h->ev_set = (h->ev_set + 1) % gndt_handle::kEvSets;
    for (auto& r : h->ev_recorded[h->ev_set]) r = false;
}

inline int grid_for(uint64_t work, int block = kBlock, int max_blocks = 256 * 8) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (uint64_t)max_blocks) b = max_blocks;
    return (int)b;
}

uint32_t pow2_ceil(uint64_t v) {
    uint64_t p = 1024;
    while (p < v && p < (1ull << 31)) p <<= 1;
    return (uint32_t)p;
}

GridParams grid_params(const gndt_handle* h) {
    GridParams g;
    g.ox = h->origin[0]; g.oy = h->origin[1]; g.oz = h->origin[2];
    g.grid_len = h->P.grid_len; g.z_len = h->P.z_len; g.slope_interval = h->P.slope_interval;
    g.demand = h->P.demand; g.min_points = h->P.min_points;
    return g;
}

void free_table(gndt_handle* h) {
    void* ptrs[] = {h->keys, h->acc, h->col_keys, h->col_first, h->aux, h->node_slot, h->col_slot_of_node,
                    h->col_cnt, h->col_head, h->node_next, h->index_of_slot, h->touch_epoch, h->col_epoch, h->touched,
                    h->touched_cols};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->keys = nullptr; h->acc = nullptr; h->col_keys = nullptr; h->col_first = nullptr; h->aux = nullptr;
    h->node_slot = nullptr; h->col_slot_of_node = nullptr; h->col_cnt = nullptr; h->col_head = nullptr; h->node_next = nullptr;
    h->index_of_slot = h->touch_epoch = h->col_epoch = h->touched = h->touched_cols = nullptr;
    h->incr_ok = false;
    h->cap = 0;
}

// A fresh, empty table of `cap` slots.  The device counters are NOT touched: the caller decides (reset vs growth).
int alloc_table(gndt_handle* h, uint32_t cap, hipStream_t s) {
    free_table(h);
    HIP_TRY(h, hipMalloc(&h->keys, (size_t)cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->acc, (size_t)cap * sizeof(NodeAcc)));
    HIP_TRY(h, hipMalloc(&h->col_keys, (size_t)cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->col_first, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_cnt, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_head, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->aux, (size_t)cap * sizeof(SlotAux)));
    HIP_TRY(h, hipMalloc(&h->node_slot, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->col_slot_of_node, (size_t)cap * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(&h->node_next, (size_t)cap * sizeof(uint32_t)));
    for (uint32_t** a : {&h->index_of_slot, &h->touch_epoch, &h->col_epoch, &h->touched, &h->touched_cols})
        HIP_TRY(h, hipMalloc(a, (size_t)cap * sizeof(uint32_t)));
    h->cap = cap;
    hipLaunchKernelGGL(k_clear_all, dim3(grid_for(cap)), dim3(kBlock), 0, s, h->keys, h->acc, h->col_keys,
                       h->col_first, h->col_cnt, h->col_head, h->touch_epoch, h->col_epoch, cap);
    HIP_TRY(h, hipGetLastError());
    h->table_dirty = false;
    return GNDT_OK;
}

int ensure_out(gndt_handle* h, uint64_t n) {
    if (n <= h->out_cap) return GNDT_OK;
    void* ptrs[] = {h->out.sx, h->out.sy, h->out.sz, h->out.count, h->out.first_idx, h->out.mean, h->out.cov,
                    h->out.rough, h->out.normal, h->out.flags};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->out = OutView{};
    h->out_cap = 0;
    uint64_t c = std::max<uint64_t>(1024, n + n / 8);
    HIP_TRY(h, hipMalloc(&h->out.sx, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.sy, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.sz, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.count, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.first_idx, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.mean, c * 12));
    HIP_TRY(h, hipMalloc(&h->out.cov, c * 24));
    HIP_TRY(h, hipMalloc(&h->out.rough, c * 4));
    HIP_TRY(h, hipMalloc(&h->out.normal, c * 12));
    HIP_TRY(h, hipMalloc(&h->out.flags, c * 4));
    h->out_cap = c;
    return GNDT_OK;
}

int ensure_stats_buffers(gndt_handle* h, uint64_t n) {
    if (n <= h->st_cap) return GNDT_OK;
    void* ptrs[] = {h->st_key, h->st_sums, h->st_count, h->st_first};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->st_key = nullptr; h->st_sums = nullptr; h->st_count = nullptr; h->st_first = nullptr; h->st_cap = 0;
    uint64_t c = std::max<uint64_t>(1024, n + n / 8);
    HIP_TRY(h, hipMalloc(&h->st_key, c * 8));
    HIP_TRY(h, hipMalloc(&h->st_sums, c * 72));
    HIP_TRY(h, hipMalloc(&h->st_count, c * 4));
    HIP_TRY(h, hipMalloc(&h->st_first, c * 4));
    h->st_cap = c;
    return GNDT_OK;
}

// read the device counters (synchronises the stream)
int fetch_counters(gndt_handle* h, hipStream_t s) {
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return GNDT_OK;
}

// slots wanted for `nodes` occupied entries (load factor <= 1/2)
uint32_t cap_for_nodes(uint64_t nodes) { return pow2_ceil(std::max<uint64_t>(2048, nodes * 2)); }

int check_ready(gndt_handle* h) {
    if (!h) return GNDT_ERR_INVALID;
    if (!h->origin_set) { h->err = "gndt_set_origin must be called first (setCloudFirst, receiver.cpp:145)"; return GNDT_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    return GNDT_OK;
}

// ---------------------------------------------------------------------------------------------
// buffers shared by both strategies' finalisation: staging rows, ordering arrays, column-first bitmap
// ---------------------------------------------------------------------------------------------
template <typename T>
int grow_buf(gndt_handle* h, T*& p, uint64_t& cap, uint64_t want) {
    if (want <= cap) return GNDT_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    HIP_TRY(h, hipMalloc(&p, want * sizeof(T)));
    cap = want;
    return GNDT_OK;
}

void free_cost(gndt_handle* h) {
    auto& c = h->cost;
    void* ptrs[] = {c.h_bits, c.pushed, c.state, c.f[0], c.f[1], c.ctab_key, c.ctab_val, c.ring, c.nbr, c.d_cc};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c.h_cc) (void)hipHostFree(c.h_cc);
    c = gndt_handle::Cost{};
}

void free_part(gndt_handle* h) {
    auto& q = h->part;
    void* ptrs[] = {q.recs, q.recs1, q.cursors, q.range_lo, q.range_hi, q.range_cap, q.hist, q.totals, q.bucket_base, q.stage, q.ord_cf, q.ord_idx, q.inv, q.row_ncol,
                    q.bitmap, q.word_weight, q.word_base, q.bsum_words, q.ncol_at, q.d_pc, q.dbg};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (q.h_pc) (void)hipHostFree(q.h_pc);
    q = gndt_handle::Part{};
}

int ensure_stage(gndt_handle* h, uint64_t nodes) {
    auto& q = h->part;
    if (nodes <= q.stage_cap) return GNDT_OK;
    void* ptrs[] = {q.stage, q.ord_cf, q.ord_idx, q.inv, q.row_ncol};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    q.stage = nullptr; q.ord_cf = q.ord_idx = q.inv = q.row_ncol = nullptr;
    q.stage_cap = 0;
    HIP_TRY(h, hipMalloc(&q.stage, nodes * sizeof(StageRow)));
    uint32_t** arrs[] = {&q.ord_cf, &q.ord_idx, &q.inv, &q.row_ncol};
    for (auto a : arrs) HIP_TRY(h, hipMalloc(a, nodes * 4));
    q.stage_cap = nodes;
    return GNDT_OK;
}

// Bucket count: ~4096 points per bucket, and few enough nodes per bucket for the LDS table
// (average load <= 0.4 of `slots`: the overflow limit of 0.78 is then ~6 sigma of the column count away).
int ensure_words(gndt_handle* h, uint64_t words) {
    auto& q = h->part;
    if (words <= q.word_cap) return GNDT_OK;
    for (uint32_t** a : {&q.bitmap, &q.word_weight, &q.word_base, &q.bsum_words, &q.ncol_at}) { if (*a) (void)hipFree(*a); *a = nullptr; }
    q.word_cap = 0;
    words += words / 4;
    HIP_TRY(h, hipMalloc(&q.bitmap, words * 4));
    HIP_TRY(h, hipMalloc(&q.word_weight, words * 4));
    HIP_TRY(h, hipMalloc(&q.word_base, words * 4));
    HIP_TRY(h, hipMalloc(&q.bsum_words, ((words + kScanChunk - 1) / kScanChunk + 1) * 4));
    HIP_TRY(h, hipMalloc(&q.ncol_at, words * 32 * 4));     // one entry per point index, touched only at column-first indices
    q.word_cap = words;
    return GNDT_OK;
}

int ensure_part_counters(gndt_handle* h) {
    auto& q = h->part;
    if (q.d_pc) return GNDT_OK;
    HIP_TRY(h, hipMalloc(&q.d_pc, sizeof(PartCounters)));
    HIP_TRY(h, hipHostMalloc(&q.h_pc, sizeof(PartCounters)));
    HIP_TRY(h, hipMemset(q.d_pc, 0, sizeof(PartCounters)));
    memset(q.h_pc, 0, sizeof(PartCounters));
    return GNDT_OK;
}

// prefix of the per-word column weights -> row of every staged node -> SoA rows (marks m0+1 .. m0+5)
int launch_order_and_emit(gndt_handle* h, uint64_t words, int m0, hipStream_t s) {
    auto& q = h->part;
    const uint32_t nbw = (uint32_t)((words + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(k_scan_reduce<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(nbw), dim3(kScanThreads), 0, s, q.word_weight, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words, q.word_base);
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 1, s);
    mark(h, m0 + 2, s);       // (the column-rank and column-scan passes of earlier versions: phases kept for the ABI, empty)
    mark(h, m0 + 3, s);
    hipLaunchKernelGGL(k_order_dest, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.ord_cf, q.ord_idx, q.bitmap, q.word_base,
                       q.ncol_at, q.inv, h->d_cnt, q.d_pc);
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 4, s);
    hipLaunchKernelGGL(k_emit_rows, dim3(grid_for(q.stage_cap)), dim3(kBlock), 0, s, q.stage, q.inv, h->out, q.row_ncol, h->d_cnt,
                       q.d_pc);
    HIP_TRY(h, hipGetLastError());
    mark(h, m0 + 5, s);
    return GNDT_OK;
}

TableView table_view(const gndt_handle* h) {
    TableView T;
    T.keys = h->keys; T.acc = h->acc; T.aux = h->aux; T.col_keys = h->col_keys; T.col_first = h->col_first;
    T.col_cnt = h->col_cnt; T.col_head = h->col_head; T.node_slot = h->node_slot; T.col_slot_of_node = h->col_slot_of_node;
    T.node_next = h->node_next; T.cap_mask = h->cap - 1;
    T.index_of_slot = h->index_of_slot; T.touch_epoch = h->touch_epoch; T.col_epoch = h->col_epoch;
    T.touched = h->touched; T.touched_cols = h->touched_cols;
    return T;
}

int do_reset(gndt_handle* h, hipStream_t s) {
    if (h->cap && h->table_dirty) {
        hipLaunchKernelGGL(k_clear_used, dim3(grid_for(h->cap / 8)), dim3(kBlock), 0, s, h->keys, h->acc, h->col_keys,
                           h->col_first, h->col_cnt, h->col_head, h->node_slot, h->col_slot_of_node, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(k_zero_counters, dim3(1), dim3(64), 0, s, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    h->table_dirty = false;
    h->results_valid = false;
    h->stream_pos = 0;
    h->nodes_bound = 0;
    h->incr_ok = false;
    return GNDT_OK;
}

// Grow the table to `new_cap` slots keeping its contents (export -> fresh table -> merge).
int grow_table(gndt_handle* h, uint32_t new_cap, hipStream_t s);

// `base_from_device`: first_idx base = the device-side stream position (incremental updates)
int do_accumulate(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint64_t first_base,
                  int base_from_device, hipStream_t s, int mark = 0) {
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (first_base + n >= 0xFFFFFFFFull) { h->err = "point index exceeds 32 bits"; return GNDT_ERR_INVALID; }
    if (n == 0) return GNDT_OK;
    const float* p = static_cast<const float*>(xyz_dev);
    const int blocks = grid_for(n, kBlock, 256 * 16);
    if (stride_bytes == 12)
        hipLaunchKernelGGL(k_accumulate<3>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, (uint32_t)first_base,
                           base_from_device, grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot,
                           h->touch_epoch, h->touched, mark, h->d_cnt);
    else
        hipLaunchKernelGGL(k_accumulate<4>, dim3(blocks), dim3(kBlock), 0, s, p, (uint64_t)n, (uint32_t)first_base,
                           base_from_device, grid_params(h), h->keys, h->acc, h->cap - 1, h->node_slot, h->index_of_slot,
                           h->touch_epoch, h->touched, mark, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    if (!mark) h->incr_ok = false;                    // nodes changed without being listed: the next finalisation redoes every column
    if (base_from_device) {
        hipLaunchKernelGGL(k_advance_stream, dim3(1), dim3(64), 0, s, h->d_cnt, (uint32_t)n);
        HIP_TRY(h, hipGetLastError());
    }
    h->table_dirty = true;
    h->results_valid = false;
    h->nodes_bound = std::min<uint64_t>(h->nodes_bound + n, h->cap);
    return GNDT_OK;
}

// columns -> labels + staging rows -> ordering -> emit.  Everything is sized on the device; nothing waits for
// the host, so accumulate + finalize can be captured in a hipGraph once the buffers exist.
int do_finalize(gndt_handle* h, hipStream_t s, bool incremental = false, uint64_t touched_bound = 0) {
    auto& q = h->part;
    int rc;
    // host-side upper bounds only: rows <= slots/2 at a healthy load; points seen so far (or the caller's hint)
    const uint64_t rows_bound = std::max<uint64_t>(1024, h->cap / 2 + 1);
    const uint64_t pts_bound = std::max<uint64_t>(std::max<uint64_t>(h->stream_pos, h->P.max_points_hint), 64);
    const uint64_t words = (pts_bound + 31) / 32 + 1;
    // The incremental form needs what the last finalisation left behind (staging rows, order keys, column order): any
    // reallocation, or anything else that touched those buffers, sends this call down the full path.
    if (!h->incr_ok || rows_bound > q.stage_cap || words > q.word_cap || !q.d_pc) incremental = false;
    if ((rc = ensure_part_counters(h))) return rc;
    if ((rc = ensure_stage(h, rows_bound))) return rc;
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    if ((rc = ensure_words(h, words))) return rc;
    const TableView T = table_view(h);
    const GridParams gp = grid_params(h);
    const ColumnOrder O{q.bitmap, q.word_weight, q.ncol_at};
    mark(h, 2, s);
    if (incremental) {
        if (words > q.words_init) {                        // the stream grew past the words the order has seen: they start empty
            HIP_TRY(h, hipMemsetAsync(q.bitmap + q.words_init, 0, (words - q.words_init) * 4, s));
            HIP_TRY(h, hipMemsetAsync(q.word_weight + q.words_init, 0, (words - q.words_init) * 4, s));
            q.words_init = words;
        }
        const uint64_t tb = std::max<uint64_t>(std::min<uint64_t>(touched_bound, rows_bound), 64);
        hipLaunchKernelGGL(k_tab_touch, dim3(grid_for(tb)), dim3(kBlock), 0, s, T, gp, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_tab_expand, dim3(grid_for(tb)), dim3(kBlock), 0, s, T, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        mark(h, 3, s);
        hipLaunchKernelGGL(k_tab_rows_touched, dim3(grid_for(4 * tb, kBlock, 4096)), dim3(kBlock), 0, s, T, gp, q.stage,
                           (uint32_t)q.stage_cap, q.ord_cf, q.ord_idx, O, (uint64_t)words, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_tab_begin, dim3(grid_for(std::max<uint64_t>(words, h->cap / 4), kBlock, 1024)), dim3(kBlock), 0, s, T,
                           h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words);
        hipLaunchKernelGGL(k_tab_columns, dim3(grid_for(rows_bound)), dim3(kBlock), 0, s, T, gp, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        mark(h, 3, s);
        hipLaunchKernelGGL(k_tab_rows, dim3(grid_for(rows_bound)), dim3(kBlock), 0, s, T, gp, q.stage, (uint32_t)q.stage_cap,
                           q.ord_cf, q.ord_idx, O, (uint64_t)words, h->d_cnt, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        q.words_init = words;
    }
    mark(h, 4, s);
    if ((rc = launch_order_and_emit(h, words, 4, s))) return rc;
    hipLaunchKernelGGL(k_tab_end, dim3(1), dim3(64), 0, s, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(q.h_pc, q.d_pc, sizeof(PartCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    h->results_valid = true;
    ++h->result_serial;
    h->last_stream = s;
    h->incr_ok = true;
    return GNDT_OK;
}

uint64_t expected_nodes_for_batch(const gndt_handle* h, uint64_t known_nodes, uint64_t n) {
    if (h->P.max_nodes_hint) return std::max<uint64_t>(h->P.max_nodes_hint, known_nodes);
    return known_nodes + n;   // worst case: every point opens a node
}

// Make room for `extra` more points (or merged nodes).  The host only tracks an upper bound of the node count;
// when that bound asks for a larger table the real count is fetched (one sync) before anything is moved.
int ensure_capacity_for(gndt_handle* h, uint64_t extra, hipStream_t s) {
    uint32_t want = cap_for_nodes(expected_nodes_for_batch(h, h->nodes_bound, extra));
    if (h->cap == 0) return alloc_table(h, want, s);
    if (want <= h->cap) return GNDT_OK;
    if (h->table_dirty) {
        int rc = fetch_counters(h, s);
        if (rc) return rc;
        h->nodes_bound = h->h_cnt->num_nodes;
        want = cap_for_nodes(expected_nodes_for_batch(h, h->nodes_bound, extra));
        if (want <= h->cap) return GNDT_OK;
    }
    return grow_table(h, want, s);
}

int grow_table(gndt_handle* h, uint32_t new_cap, hipStream_t s) {
    if (!h->table_dirty) return alloc_table(h, new_cap, s);
    // export the current contents (the node list is always valid), rebuild, merge back
    int rc = fetch_counters(h, s);
    if (rc) return rc;
    const uint32_t C = h->h_cnt->num_nodes;
    rc = ensure_stats_buffers(h, C);
    if (rc) return rc;
    if (C) {
        hipLaunchKernelGGL(k_stats_export, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->node_slot, h->d_cnt,
                           h->st_key, h->st_sums, h->st_count, h->st_first);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    rc = alloc_table(h, new_cap, s);
    if (rc) return rc;
    // the new table is empty: node list restarts, nobody owns a column entry yet (stream position is kept)
    HIP_TRY(h, hipMemsetAsync(&h->d_cnt->num_nodes, 0, sizeof(uint32_t), s));
    HIP_TRY(h, hipMemsetAsync(&h->d_cnt->prev_nodes, 0, sizeof(uint32_t), s));
    if (C) {
        hipLaunchKernelGGL(k_stats_merge, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->cap - 1, h->node_slot,
                           h->index_of_slot, h->st_key, h->st_sums, h->st_count, h->st_first, (uint64_t)C, h->d_cnt);
        HIP_TRY(h, hipGetLastError());
        h->table_dirty = true;
    }
    h->nodes_bound = C;
    return GNDT_OK;
}

int stage_host_input(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes, hipStream_t s) {
    const size_t bytes = n * stride_bytes;
    if (bytes > h->stage_bytes) {
        if (h->stage) (void)hipFree(h->stage);
        h->stage = nullptr; h->stage_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->stage, bytes + 64));
        h->stage_bytes = bytes;
    }
    if (bytes) HIP_TRY(h, hipMemcpyAsync(h->stage, xyz_host, bytes, hipMemcpyHostToDevice, s));
    return GNDT_OK;
}

// ---------------------------------------------------------------------------------------------
// strategy PARTITION (gndt_partition.hpp): the build
// ---------------------------------------------------------------------------------------------
// Bucket count for `nodes` expected nodes.  Few, large buckets are better for both the scatter (longer runs per
// workgroup and bucket) and the bucket kernel (fixed costs per bucket): as many points per bucket as two chunks of
// the bucket kernel take (2800 leaves room for the spread of a hash partition), unless the LDS node table says
// otherwise: average load <= 0.5 of `slots` against an estimate that already carries a 20 % margin, i.e. ~0.42 of
// the slots really used, ~5 sigma of the column count below the overflow limit of 0.78.  (An overflow is not an
// error: the build is re-run with the larger table / more buckets.)
uint64_t buckets_for(uint64_t n, uint64_t nodes, int slots) {
    static const int load_pct = getenv("GNDT_BUCKET_LOAD") ? atoi(getenv("GNDT_BUCKET_LOAD")) : 50;       // tuning knobs
    static const int pts_target = getenv("GNDT_BUCKET_POINTS") ? atoi(getenv("GNDT_BUCKET_POINTS")) : 0;
    if (pts_target) return std::max<uint64_t>(n / (uint64_t)pts_target, 16);
    const uint64_t per_bucket = slots >= 1024 ? 6400 : 2800;
    const uint64_t want = std::max<uint64_t>(n / per_bucket, (nodes * 100) / ((uint64_t)slots * load_pct));
    return std::max<uint64_t>(want, 16);
}
constexpr uint64_t kMaxBuckets = 32768;   // 4-byte LDS cursor per bucket in the partition passes

// One attempt of the PARTITION build: every launch plus the asynchronous read-back of the counters and overflow
// flags; no host wait.  Returns GNDT_OK, an error, or -1 when the partition path cannot hold this input.
int partition_launch(gndt_handle* h, gndt_handle::Pending& P) {
    auto& q = h->part;
    const size_t n = P.n, stride_bytes = P.stride;
    hipStream_t s = P.s;
    const int attempt = P.attempt;
    uint64_t& nodes_est = P.nodes_est;
    uint64_t& stage_want = P.stage_want;
    static const int bt = getenv("GNDT_BUCKET_THREADS") ? atoi(getenv("GNDT_BUCKET_THREADS")) : 512;
    static const int env_slots = getenv("GNDT_BUCKET_SLOTS") ? atoi(getenv("GNDT_BUCKET_SLOTS")) : 0;
    static const int part_wgs = getenv("GNDT_PART_WGS") ? atoi(getenv("GNDT_PART_WGS")) : 256;
    const uint32_t nwg = (uint32_t)std::min<uint64_t>((uint64_t)part_wgs, std::max<uint64_t>(1, n / 8192));
    const uint64_t words = (n + 31) / 32 + 1;
    int rc;
    const GridParams gp = grid_params(h);
    const float* p = static_cast<const float*>(P.xyz);
    // table size and bucket count for this attempt: 512-slot tables unless that needs too many buckets
    // attempt 0: 512-slot tables; an overflow first doubles the table (same estimate), then raises the estimate
    int bslots = env_slots ? env_slots : (attempt == 0 ? 512 : 1024);
    uint64_t Bw = buckets_for(n, nodes_est, bslots);
    // Two-level partition (no counting passes) for large builds; the exact single-level counting partition for small
    // ones, when asked for (GNDT_STRATEGY_PARTITION_EXACT), and after a region overflowed once on this handle.
    static const int env_two = getenv("GNDT_TWO_LEVEL") ? atoi(getenv("GNDT_TWO_LEVEL")) : -1;
    bool two = h->P.strategy != GNDT_STRATEGY_PARTITION_EXACT && q.two_level_ok &&
               (n >= (1u << 20) || h->P.strategy == GNDT_STRATEGY_PARTITION_TWO_LEVEL);
    if (env_two == 0 || n == 0) two = false;
    if (two && Bw > (uint64_t)kMaxFan * kMaxFan) {         // more buckets than two levels address: larger tables, fewer buckets
        if (!env_slots) { bslots = 1024; Bw = buckets_for(n, nodes_est, bslots); }
        if (Bw > (uint64_t)kMaxFan * kMaxFan) two = false;
    }
    if (!two) {
        if (!env_slots && Bw > kMaxBuckets) { bslots = 1024; Bw = buckets_for(n, nodes_est, bslots); }
        if (Bw > kMaxBuckets) return -1;                   // too many nodes for one partition level: atomic path
    }
    const uint32_t B = (uint32_t)Bw;
    P.two_level = two;
    h->last_strategy = two ? GNDT_STRATEGY_PARTITION : GNDT_STRATEGY_PARTITION_EXACT;
    stage_want = std::max<uint64_t>(stage_want, nodes_est + nodes_est / 8);
    if (P.stats_only) {
        if ((rc = ensure_stats_buffers(h, stage_want))) return rc;
    } else {
        if ((rc = ensure_stage(h, stage_want))) return rc;
        if ((rc = ensure_out(h, q.stage_cap))) return rc;
    }
    const uint32_t* range_lo = nullptr;
    const uint32_t* range_hi = nullptr;
    const float4* bucket_recs = nullptr;
    if (two) {
        uint32_t F2_shift = 1;                             // fan-out ~ sqrt(B) per level, F2 a power of two, both <= kMaxFan
        while (F2_shift < 9 && (1ull << (2 * F2_shift)) < B) ++F2_shift;
        const uint32_t F2 = 1u << F2_shift;
        const uint32_t F1 = (B + F2 - 1) / F2;
        // Level-1 regions are large and hash-balanced: a fixed capacity of 2x the mean (or 1.25x the fullest one an
        // earlier build on this handle saw).  The buckets' regions are laid out on the device from a 1-in-64 sample
        // level 1 takes (k_part2_layout: 2x the estimate + 2048 each), so LiDAR clouds' hot columns get the room they
        // need and the records take 2 n + 2048 B slots in all.  If a region overflows all the same, the build is
        // re-run; a second failure sends this handle to the exact counting partition.
        static const int env_rep = getenv("GNDT_L1_REP") ? atoi(getenv("GNDT_L1_REP")) : 1;   // measured: 1 is best at 4096-point tiles
        const uint32_t R = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)env_rep, kMaxFan / F1));   // sub-regions per coarse region
        const uint32_t V = F1 * R;
        constexpr uint64_t kTile1 = (uint64_t)kTileThreads * kTilePer1, kTile2 = (uint64_t)kTileThreads * kTilePer2;
        const double r1 = std::max(2.0, q.fill1_ratio * 1.25);
        const uint64_t cap1w = (uint64_t)(r1 * (double)(n / V)) + 2 * kTile1;
        const uint64_t recs_want = 2 * (uint64_t)n + n / 8 + 2048ull * B + 4096;     // 2 n + 2048 B, and sampling slack
        if ((uint64_t)V * cap1w > 4 * (uint64_t)n + (1u << 24) || recs_want >= 0xF0000000ull || q.two_level_failures >= 2) {
            q.two_level_ok = false;
            return partition_launch(h, P);                 // (re-enters on the exact path)
        }
        const uint32_t cap1 = (uint32_t)cap1w;
        P.mean1 = (double)(n / V);
        if ((rc = grow_buf(h, q.recs1, q.rec1_cap, (uint64_t)V * cap1))) return rc;
        if ((rc = grow_buf(h, q.recs, q.rec_cap, recs_want))) return rc;
        if (B > q.cur_cap) {
            for (uint32_t** a : {&q.cursors, &q.range_lo, &q.range_hi, &q.range_cap}) { if (*a) (void)hipFree(*a); *a = nullptr; }
            q.cur_cap = 0;
            const uint64_t c = (uint64_t)B + B / 4;
            HIP_TRY(h, hipMalloc(&q.cursors, ((size_t)kMaxFan + 2 * c) * 4));  // [kMaxFan] level 1, [c] level 2, [c] samples
            HIP_TRY(h, hipMalloc(&q.range_lo, c * 4));
            HIP_TRY(h, hipMalloc(&q.range_hi, c * 4));
            HIP_TRY(h, hipMalloc(&q.range_cap, c * 4));
            q.cur_cap = c;
        }
        uint32_t* cursor1 = q.cursors;
        uint32_t* cursor2 = q.cursors + kMaxFan;
        uint32_t* est2 = cursor2 + B;
        mark(h, 0, s);
        if (h->table_dirty) { if ((rc = do_reset(h, s))) return rc; }
        h->results_valid = false;
        hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                           q.cursors, (uint32_t)(kMaxFan + 2 * B));
        HIP_TRY(h, hipGetLastError());
        mark(h, 1, s);
        const uint32_t tiles1 = (uint32_t)((n + kTile1 - 1) / kTile1);
        static const uint32_t l1_wgs = getenv("GNDT_L1_WGS") ? (uint32_t)atoi(getenv("GNDT_L1_WGS")) : 1024u;   // persistent workgroups (2 resident per CU)
        const bool wide = std::max(V, F2) > 256;           // LDS arrays for a fan-out of 512 (fewer resident tiles) only when needed
        const dim3 g1(std::min<uint32_t>(tiles1, l1_wgs)), g2((uint32_t)((cap1 + kTile2 - 1) / kTile2), V);
#define GNDT_L1(SF_, FAN_)                                                                                                  \
    hipLaunchKernelGGL((k_part2_level1<SF_, FAN_>), g1, dim3(kTileThreads), 0, s, p, (uint64_t)n, P.first_base, gp, B, F1, F2_shift, \
                       R, cursor1, cap1, est2, q.recs1, h->d_cnt, q.d_pc)
        if (stride_bytes == 12) { if (wide) GNDT_L1(3, 512); else GNDT_L1(3, 256); }
        else { if (wide) GNDT_L1(4, 512); else GNDT_L1(4, 256); }
#undef GNDT_L1
        HIP_TRY(h, hipGetLastError());
        mark(h, 2, s);
        hipLaunchKernelGGL(k_part2_layout, dim3(1), dim3(1024), 0, s, est2, B, q.range_lo, q.range_cap, (uint64_t)q.rec_cap, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        mark(h, 3, s);
        if (wide)
            hipLaunchKernelGGL(k_part2_level2<512>, g2, dim3(kTileThreads), 0, s, q.recs1, cursor1, cap1, R, gp, B, F2, cursor2,
                               q.range_lo, q.range_cap, q.recs, q.d_pc);
        else
            hipLaunchKernelGGL(k_part2_level2<256>, g2, dim3(kTileThreads), 0, s, q.recs1, cursor1, cap1, R, gp, B, F2, cursor2,
                               q.range_lo, q.range_cap, q.recs, q.d_pc);
        hipLaunchKernelGGL(k_part2_ranges, dim3(grid_for(B, 256, 64)), dim3(256), 0, s, cursor1, V, cursor2, q.range_cap, B, q.range_lo,
                           q.range_hi, q.d_pc);
        HIP_TRY(h, hipGetLastError());
        mark(h, 4, s);
        range_lo = q.range_lo; range_hi = q.range_hi;
    } else {
    if ((rc = grow_buf(h, q.recs, q.rec_cap, n))) return rc;
    if ((rc = grow_buf(h, q.hist, q.hist_cap, (uint64_t)nwg * B))) return rc;
    if (B > q.bucket_cap) {
        if (q.totals) (void)hipFree(q.totals);
        if (q.bucket_base) (void)hipFree(q.bucket_base);
        q.totals = q.bucket_base = nullptr; q.bucket_cap = 0;
        HIP_TRY(h, hipMalloc(&q.totals, (size_t)B * 4));
        HIP_TRY(h, hipMalloc(&q.bucket_base, ((size_t)B + 1) * 4));
        q.bucket_cap = B;
    }
    mark(h, 0, s);
    // the HBM node table is not used by this strategy, but a previous atomic build may sit in it
    if (h->table_dirty) { if ((rc = do_reset(h, s))) return rc; }
    h->results_valid = false;
    hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                       (uint32_t*)nullptr, 0u);
    HIP_TRY(h, hipGetLastError());
    mark(h, 1, s);
    const size_t lds = (size_t)B * 4;
    if (lds > 48 * 1024) {   // beyond the default dynamic-LDS limit the kernels must be told (gfx950: 160 KiB/CU)
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_hist<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_hist<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_scatter<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_part_scatter<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (stride_bytes == 12)
        hipLaunchKernelGGL(k_part_hist<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, B, q.hist, h->d_cnt);
    else
        hipLaunchKernelGGL(k_part_hist<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, gp, B, q.hist, h->d_cnt);
    HIP_TRY(h, hipGetLastError());
    mark(h, 2, s);
    hipLaunchKernelGGL(k_part_offsets, dim3((B + 31) / 32), dim3(256), 0, s, q.hist, q.totals, B, nwg);
    HIP_TRY(h, hipGetLastError());
    mark(h, 3, s);
    if (stride_bytes == 12)
        hipLaunchKernelGGL(k_part_scatter<3>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, P.first_base, gp, B, q.hist,
                           q.totals, q.bucket_base, q.recs);
    else
        hipLaunchKernelGGL(k_part_scatter<4>, dim3(nwg), dim3(kPartThreads), lds, s, p, (uint64_t)n, P.first_base, gp, B, q.hist,
                           q.totals, q.bucket_base, q.recs);
    HIP_TRY(h, hipGetLastError());
    mark(h, 4, s);
    range_lo = q.bucket_base; range_hi = q.bucket_base + 1;
    }
    bucket_recs = q.recs;
    if (getenv("GNDT_STAMPS") && q.dbg_buckets < B) {
        if (q.dbg) (void)hipFree(q.dbg);
        q.dbg = nullptr; q.dbg_buckets = 0;
        HIP_TRY(h, hipMalloc(&q.dbg, (size_t)B * 16 * sizeof(unsigned long long)));
        HIP_TRY(h, hipMemset(q.dbg, 0, (size_t)B * 16 * sizeof(unsigned long long)));
        q.dbg_buckets = B;
    }
    q.last_buckets = B;
    // one workgroup per bucket by default: persistent workgroups (GNDT_BUCKET_WGS=512) measured 8 % slower, the
    // hardware's dynamic workgroup scheduling balances uneven buckets better than a static stride
    static const uint32_t bucket_wgs = getenv("GNDT_BUCKET_WGS") ? (uint32_t)atoi(getenv("GNDT_BUCKET_WGS")) : 0xFFFFFFFFu;
    {
#define GNDT_LAUNCH_BUCKET2(T_, H_, CH_, S_)                                                                           \
    hipLaunchKernelGGL((k_bucket_build2<T_, H_, CH_, S_>), dim3(std::min<uint32_t>(B, bucket_wgs)), dim3(T_), 0, s, bucket_recs, range_lo, range_hi, B, gp, q.stage, \
                       (uint32_t)(S_ ? h->st_cap : q.stage_cap), q.ord_cf, q.ord_idx, ColumnOrder{q.bitmap, q.word_weight, q.ncol_at}, h->d_cnt,   \
                       q.d_pc, q.dbg, StatsOut{h->st_key, h->st_sums, h->st_count, h->st_first})
            if (P.stats_only) {
                if (bslots == 1024) GNDT_LAUNCH_BUCKET2(512, 1024, 3584, true);
                else GNDT_LAUNCH_BUCKET2(512, 512, 1536, true);
            }
            else if (bslots == 1024 && bt == 1024) GNDT_LAUNCH_BUCKET2(1024, 1024, 3072, false);
            else if (bslots == 1024) GNDT_LAUNCH_BUCKET2(512, 1024, 3584, false);
            else if (bslots == 512 && bt == 256) GNDT_LAUNCH_BUCKET2(256, 512, 1792, false);
            else if (bslots == 256) GNDT_LAUNCH_BUCKET2(256, 256, 1024, false);
            else GNDT_LAUNCH_BUCKET2(512, 512, 1536, false);
#undef GNDT_LAUNCH_BUCKET2
    }
    HIP_TRY(h, hipGetLastError());
    mark(h, 5, s);
    if (!P.stats_only && (rc = launch_order_and_emit(h, words, 5, s))) return rc;
    HIP_TRY(h, hipMemcpyAsync(q.h_pc, q.d_pc, sizeof(PartCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    P.bslots = bslots;
    return GNDT_OK;
}

int build_atomic(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s);

// Start a PARTITION build (attempt 0) and leave it pending.
int partition_begin(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s) {
    auto& q = h->part;
    if (stride_bytes != 12 && stride_bytes != 16) { h->err = "stride_bytes must be 12 or 16"; return GNDT_ERR_INVALID; }
    if (n >= 0x7FFFFFFFull) return -1;   // bit 31 of the record index word carries the weight flag: atomic path beyond 2^31 points
    int rc;
    if ((rc = ensure_words(h, (n + 31) / 32 + 1))) return rc;
    if ((rc = ensure_part_counters(h))) return rc;
    auto& P = h->pending;
    P = gndt_handle::Pending{};
    P.xyz = xyz_dev; P.n = n; P.stride = stride_bytes; P.s = s; P.attempt = 0;
    // expected node count: the caller's hint, else what the previous build of this handle found, else n/4
    P.nodes_est = h->P.max_nodes_hint ? h->P.max_nodes_hint : (q.nodes_learned ? q.nodes_learned : std::max<uint64_t>(n / 4, 1024));
    P.stage_want = std::max<uint64_t>(q.stage_cap, h->P.max_nodes_hint ? h->P.max_nodes_hint + h->P.max_nodes_hint / 8
                                                                       : std::max<uint64_t>(4096, n / 4));
    const int prev_strategy = h->last_strategy;
    h->last_strategy = GNDT_STRATEGY_PARTITION;
    rc = partition_launch(h, P);
    if (rc) { h->last_strategy = prev_strategy; return rc; }
    P.active = true;
    h->results_valid = false;
    h->map_in_table = false;
    h->stream_pos = n;
    h->last_stream = s;
    return GNDT_OK;
}

// Wait for the pending build and look at its flags; re-run it with more room while they ask for it (the input
// must still be valid: it is the caller's until gndt_sync / gndt_export returns).
int partition_resolve(gndt_handle* h) {
    auto& P = h->pending;
    if (!P.active) return GNDT_OK;
    auto& q = h->part;
    static const int env_slots = getenv("GNDT_BUCKET_SLOTS") ? atoi(getenv("GNDT_BUCKET_SLOTS")) : 0;
    int rc = GNDT_OK;
    for (;;) {
        if (hipStreamSynchronize(P.s) != hipSuccess) { P.active = false; h->err = "hipStreamSynchronize failed"; return GNDT_ERR_HIP; }
        bool again = false;
        if (P.two_level && P.mean1 > 0) {                      // remember how uneven the level-1 regions of the latest cloud were
            q.fill1_ratio = q.h_pc->max_fill1 / P.mean1;
            if (getenv("GNDT_VERBOSE"))
                fprintf(stderr, "[gndt] two-level partition: fullest level-1 region %.2fx the mean, overflow %u\n",
                        q.h_pc->max_fill1 / P.mean1, q.h_pc->part_overflow);
        }
        if (q.h_pc->part_overflow) {                           // a region of the two-level partition was too small: same table
            ++q.two_level_failures;                            // size and estimate again (level-1 regions sized from the fullest
            --P.attempt;                                       // one seen; after two failures the exact counting partition)
            again = true;
        } else if (q.h_pc->lds_overflow) {                            // some bucket holds too many nodes for its LDS table:
            if (P.attempt >= 1 || env_slots) P.nodes_est *= 2;   // (attempt 0 -> 1 only switches to the 1024-slot table)
            again = true;
        } else if (q.h_pc->stage_overflow) {                   // num_nodes kept counting: it is the true total
            P.stage_want = (uint64_t)h->h_cnt->num_nodes + h->h_cnt->num_nodes / 8 + 1024;
            again = true;
        }
        if (!again) {
            q.nodes_learned = (uint64_t)h->h_cnt->num_nodes + h->h_cnt->num_nodes / 5;
            q.good_slots = P.bslots; q.good_est = P.nodes_est; q.good_n = P.n;
            if (!P.stats_only) {
                h->results_valid = true;
                ++h->result_serial;
            }
            h->table_dirty = false;
            P.active = false;
            return GNDT_OK;
        }
        rc = -1;
        if (++P.attempt < 5) rc = partition_launch(h, P);
        if (rc == GNDT_OK) continue;
        P.active = false;
        if (rc != -1 || P.stats_only) return rc;   // (a statistics-only run reports -1: its caller falls back)
        // does not fit the LDS-resident pipeline (too many nodes per bucket): same result via the atomic path
        return build_atomic(h, P.xyz, P.n, P.stride, P.s);
    }
}

}  // namespace

extern "C" {

const char* gndt_last_error(const gndt_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gndt_create(const gndt_params* params, gndt_handle** out) {
    if (!params || !out) { g_create_error = "null argument"; return GNDT_ERR_INVALID; }
    *out = nullptr;
    if (!(params->grid_len > 0.f) || !(params->z_len > 0.f) || params->min_points < 1 ||
        (params->demand != GNDT_DEMAND_SLOPE && params->demand != GNDT_DEMAND_TRUE)) {
        g_create_error = "invalid gndt_params (grid_len/z_len must be > 0, demand 0|1, min_points >= 1)";
        return GNDT_ERR_INVALID;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (libgndt has no CPU path)";
        return GNDT_ERR_NO_DEVICE;
    }
    if (params->device_id < 0 || params->device_id >= ndev) { g_create_error = "device_id out of range"; return GNDT_ERR_INVALID; }
    gndt_handle* h = new (std::nothrow) gndt_handle;
    if (!h) { g_create_error = "out of host memory"; return GNDT_ERR_NOMEM; }
    h->P = *params;
    h->device = params->device_id;
    auto fail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        gndt_destroy(h);
        return GNDT_ERR_HIP;
    };
    if ((e = hipSetDevice(h->device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipMalloc(&h->d_cnt, sizeof(Counters))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipHostMalloc(&h->h_cnt, sizeof(Counters))) != hipSuccess) return fail("hipHostMalloc", e);
    if ((e = hipMemset(h->d_cnt, 0, sizeof(Counters))) != hipSuccess) return fail("hipMemset", e);
    memset(h->h_cnt, 0, sizeof(Counters));
    h->last_stream = h->own_stream;
    if (params->max_nodes_hint) {
        int rc = alloc_table(h, cap_for_nodes(params->max_nodes_hint), h->own_stream);
        if (rc) { g_create_error = h->err; gndt_destroy(h); return rc; }
        if ((e = hipStreamSynchronize(h->own_stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    }
    *out = h;
    return GNDT_OK;
}

void gndt_destroy(gndt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    free_table(h);
    free_part(h);
    free_cost(h);
    void* ptrs[] = {h->out.sx, h->out.sy, h->out.sz, h->out.count, h->out.first_idx, h->out.mean, h->out.cov,
                    h->out.rough, h->out.normal, h->out.flags, h->st_key, h->st_sums, h->st_count, h->st_first,
                    h->stage, h->d_cnt, h->packed, h->d_nvalid};
    if (h->h_nvalid) (void)hipHostFree(h->h_nvalid);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& set : h->ev)
        for (auto& e : set)
            if (e) (void)hipEventDestroy(e);
    if (h->h_cnt) (void)hipHostFree(h->h_cnt);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int gndt_set_origin(gndt_handle* h, const float origin_xyz[3]) {
    if (!h || !origin_xyz) return GNDT_ERR_INVALID;
    if (h->table_dirty) { h->err = "origin cannot change while the map holds points (call gndt_reset)"; return GNDT_ERR_INVALID; }
    memcpy(h->origin, origin_xyz, 3 * sizeof(float));
    h->origin_set = true;
    return GNDT_OK;
}

int gndt_get_origin(const gndt_handle* h, float origin_xyz[3]) {
    if (!h || !origin_xyz || !h->origin_set) return GNDT_ERR_INVALID;
    memcpy(origin_xyz, h->origin, 3 * sizeof(float));
    return GNDT_OK;
}

int gndt_reset(gndt_handle* h, void* hip_stream) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    h->pending.active = false;
    h->map_in_table = true;
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    return do_reset(h, s);
}

int gndt_accumulate_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes,
                           uint64_t first_idx_base, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state: create the handle "
                 "with strategy = GNDT_STRATEGY_ATOMIC for incremental updates, or call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    rc = ensure_capacity_for(h, n, s);
    if (rc) return rc;
    mark(h, 1, s);
    rc = do_accumulate(h, xyz_dev, n, stride_bytes, first_idx_base, 0, s);
    if (rc) return rc;
    mark(h, 2, s);
    h->stream_pos = std::max<uint64_t>(h->stream_pos, first_idx_base + n);
    hipLaunchKernelGGL(k_raise_stream, dim3(1), dim3(64), 0, s, h->d_cnt, (uint32_t)h->stream_pos);
    HIP_TRY(h, hipGetLastError());
    return GNDT_OK;
}

int gndt_finalize_device(gndt_handle* h, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    if (!h->map_in_table) {
        h->err = "nothing accumulated in the node table: the current map was built by a PARTITION strategy (call gndt_reset, "
                 "then gndt_accumulate_device / gndt_stats_merge_device)";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    if (h->cap == 0) {
        rc = alloc_table(h, cap_for_nodes(1024), s);
        if (rc) return rc;
    }
    return do_finalize(h, s);
}

}  // extern "C"

namespace {
// strategy ATOMIC from empty; waits for the result (the retry on a full table needs the device-side flags)
int build_atomic(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, hipStream_t s) {
    int rc;
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    h->map_in_table = true;
    uint64_t expect = h->P.max_nodes_hint ? h->P.max_nodes_hint : std::max<uint64_t>(1024, n / 4);
    for (int attempt = 0; attempt < 8; ++attempt) {
        const uint32_t want = cap_for_nodes(expect);
        if (h->cap < want) { rc = alloc_table(h, want, s); if (rc) return rc; }
        mark(h, 0, s);
        rc = do_reset(h, s);
        if (rc) return rc;
        mark(h, 1, s);
        rc = do_accumulate(h, xyz_dev, n, stride_bytes, 0, 0, s);
        if (rc) return rc;
        h->stream_pos = n;
        hipLaunchKernelGGL(k_raise_stream, dim3(1), dim3(64), 0, s, h->d_cnt, (uint32_t)n);
        HIP_TRY(h, hipGetLastError());
        rc = do_finalize(h, s);
        if (rc) return rc;
        // a build returns with its results ready: wait once and look at the device-side flags
        HIP_TRY(h, hipStreamSynchronize(s));
        if (!h->h_cnt->err_table_full && !h->part.h_pc->stage_overflow) {
            // what a later PARTITION build of a similar cloud should expect (a first build without a hint guesses n / 4)
            h->part.nodes_learned = (uint64_t)h->h_cnt->num_nodes + h->h_cnt->num_nodes / 5;
            return GNDT_OK;
        }
        // table (or staging) overflowed: the build starts from empty, so simply redo it in a larger table
        expect = (uint64_t)h->cap * 2;   // cap_for_nodes doubles again -> 4x slots
        if (expect > (1ull << 30)) break;
    }
    return GNDT_ERR_CAPACITY;
}
}  // namespace

extern "C" {

int gndt_build_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->pending.active = false;          // a build still pending is being replaced: nobody will ask for its result
    h->last_stream = s;
    next_event_set(h);
    int strategy = h->P.strategy;
    if (strategy == GNDT_STRATEGY_AUTO) strategy = (n >= (1u << 16)) ? GNDT_STRATEGY_PARTITION : GNDT_STRATEGY_ATOMIC;
    if (strategy == GNDT_STRATEGY_PARTITION || strategy == GNDT_STRATEGY_PARTITION_EXACT || strategy == GNDT_STRATEGY_PARTITION_TWO_LEVEL) {
        // launched, not awaited: gndt_sync / gndt_export* (or whatever needs the result next) waits, checks the
        // overflow flags and re-runs with more room if needed.  xyz_dev stays the caller's until then.
        rc = partition_begin(h, xyz_dev, n, stride_bytes, s);
        if (rc != -1) return rc;
        // does not fit the LDS-resident pipeline: same result via the atomic path
    }
    return build_atomic(h, xyz_dev, n, stride_bytes, s);
}

int gndt_update_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_dev && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state: create the handle "
                 "with strategy = GNDT_STRATEGY_ATOMIC for incremental updates, or call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    next_event_set(h);
    h->last_strategy = GNDT_STRATEGY_ATOMIC;
    // Host side: only upper bounds, so that buffers exist (allocation happens outside any graph capture: run one
    // frame eagerly first, or give max_nodes_hint / max_points_hint).  The first_idx base is the DEVICE-side
    // stream position, which k_advance_stream bumps, so a captured update can be replayed frame after frame.
    rc = ensure_capacity_for(h, n, s);
    if (rc) return rc;
    mark(h, 1, s);
    // The frame's points list the nodes they touch; if the staging rows and the column order of the last finalisation
    // are still in place, only the columns holding a touched node are relabelled (the ordering and the emit pass
    // still cover the whole map: rows move when a column in front of them grows).
    const bool incr = h->incr_ok;
    rc = do_accumulate(h, xyz_dev, n, stride_bytes, h->stream_pos, 1, s, incr ? 1 : 0);
    if (rc) return rc;
    h->stream_pos += n;
    return do_finalize(h, s, incr, n);
}

int gndt_build(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    rc = stage_host_input(h, xyz_host, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    rc = gndt_build_device(h, h->stage, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

int gndt_update(gndt_handle* h, const void* xyz_host, size_t n, size_t stride_bytes) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!xyz_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    rc = stage_host_input(h, xyz_host, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    rc = gndt_update_device(h, h->stage, n, stride_bytes, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

int gndt_sync(gndt_handle* h, uint64_t* num_nodes, uint64_t* num_columns, uint64_t* num_slopes) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    { const int prc = partition_resolve(h); if (prc) return prc; }
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    if (h->results_valid) {
        h->res_nodes = h->h_cnt->num_nodes;
        h->res_columns = h->h_cnt->num_columns;
        h->res_slopes = h->h_cnt->num_slopes;
    }
    if (num_nodes) *num_nodes = h->res_nodes;
    if (num_columns) *num_columns = h->res_columns;
    if (num_slopes) *num_slopes = h->res_slopes;
    if (h->h_cnt->err_key_range) {
        h->err = std::to_string(h->h_cnt->err_key_range) +
                 " point(s) outside the key range (|nx|,|ny| <= 65535: countMorton wraps beyond, Stopwatch.h:102-110)";
        return GNDT_ERR_KEY_RANGE;
    }
    if (h->h_cnt->err_table_full) { h->err = "node table full: raise gndt_params.max_nodes_hint"; return GNDT_ERR_CAPACITY; }
    if (h->results_valid && h->part.h_pc && h->part.h_pc->stage_overflow) {
        h->err = "more nodes than result rows: raise gndt_params.max_nodes_hint";
        return GNDT_ERR_CAPACITY;
    }
    if (h->results_valid && h->part.h_pc && h->part.h_pc->index_overflow) {
        h->err = "point stream ran past the column-order bitmap (replayed graph?): raise gndt_params.max_points_hint";
        return GNDT_ERR_CAPACITY;
    }
    return GNDT_OK;
}

int gndt_export_device(gndt_handle* h, gndt_cells* out) {
    if (!h || !out) return GNDT_ERR_INVALID;
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to export"; return GNDT_ERR_INVALID; }
    int rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    out->num_nodes = h->res_nodes; out->num_columns = h->res_columns; out->num_slopes = h->res_slopes;
    out->sx = h->out.sx; out->sy = h->out.sy; out->sz = h->out.sz;
    out->count = h->out.count; out->first_idx = h->out.first_idx;
    out->mean = h->out.mean; out->cov = h->out.cov; out->rough = h->out.rough; out->normal = h->out.normal;
    out->flags = h->out.flags;
    return GNDT_OK;
}

int gndt_export(gndt_handle* h, gndt_cells* o) {
    if (!h || !o) return GNDT_ERR_INVALID;
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to export"; return GNDT_ERR_INVALID; }
    int rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    const uint64_t n = h->res_nodes;
    o->num_nodes = n; o->num_columns = h->res_columns; o->num_slopes = h->res_slopes;
    struct { void* dst; const void* src; size_t elem; } copies[] = {
        {o->sx, h->out.sx, 4}, {o->sy, h->out.sy, 4}, {o->sz, h->out.sz, 4}, {o->count, h->out.count, 4},
        {o->first_idx, h->out.first_idx, 4}, {o->mean, h->out.mean, 12}, {o->cov, h->out.cov, 24},
        {o->rough, h->out.rough, 4}, {o->normal, h->out.normal, 12}, {o->flags, h->out.flags, 4}};
    for (auto& c : copies)
        if (c.dst && n) HIP_TRY(h, hipMemcpy(c.dst, c.src, n * c.elem, hipMemcpyDeviceToHost));
    return GNDT_OK;
}

// ---------------------------------------------------------------------------------------------
// cost-map flood (TwoDmap::computeCost, include/map2D.h:1285-1397) over the finished grid
// ---------------------------------------------------------------------------------------------
constexpr int kCostBlocks = 128, kCostThreads = 64, kCostBatch = 32;

int gndt_compute_cost(gndt_handle* h, const float goal_xyz[3], const gndt_robot* robot, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!goal_xyz) { h->err = "null goal"; return GNDT_ERR_INVALID; }
    { const int prc = partition_resolve(h); if (prc) return prc; }
    if (!h->results_valid) { h->err = "no finished build to flood (computeCost runs after create2DMap, receiver.cpp:160, 171)"; return GNDT_ERR_INVALID; }
    rc = gndt_sync(h, nullptr, nullptr, nullptr);
    if (rc) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    auto& c = h->cost;
    c.serial = 0;
    const uint64_t n = h->res_nodes, K = h->res_columns;
    if (!c.d_cc) {
        HIP_TRY(h, hipMalloc(&c.d_cc, sizeof(CostCounters)));
        HIP_TRY(h, hipHostMalloc(&c.h_cc, sizeof(CostCounters)));
        HIP_TRY(h, hipMalloc(&c.ring, (size_t)kCostBlocks * kCostThreads * kRingCap * sizeof(uint32_t)));
    }
    if (n > c.node_cap) {
        for (uint32_t** a : {&c.h_bits, &c.pushed, &c.state, &c.f[0], &c.f[1], &c.nbr}) { if (*a) (void)hipFree(*a); *a = nullptr; }
        c.node_cap = 0;
        const uint64_t cap = std::max<uint64_t>(1024, n + n / 8);
        for (uint32_t** a : {&c.h_bits, &c.pushed, &c.state, &c.f[0], &c.f[1]}) HIP_TRY(h, hipMalloc(a, cap * 4));
        HIP_TRY(h, hipMalloc(&c.nbr, cap * 16));
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
    hipLaunchKernelGGL(k_cost_clear, dim3(grid_for(std::max<uint64_t>(n, c.ctab_size))), dim3(256), 0, s, c.h_bits, c.pushed,
                       c.state, (uint32_t)n, c.ctab_key, c.ctab_size, c.d_cc);
    if (K)
        hipLaunchKernelGGL(k_cost_columns, dim3(grid_for(n)), dim3(256), 0, s, h->out.sx, h->out.sy, h->part.row_ncol,
                           (uint32_t)n, c.ctab_key, c.ctab_val, c.ctab_size - 1, c.d_cc);
    if (K) {
        hipLaunchKernelGGL(k_cost_neighbours, dim3(grid_for(4 * n)), dim3(256), 0, s, V, (uint32_t)n, c.nbr);   // (probes: V.nbr is null)
        V.nbr = c.nbr;
    }
    if (gk.ok && K)
        hipLaunchKernelGGL(k_cost_goal, dim3(1), dim3(64), 0, s, V, gk.sx, gk.sy, gk.sz, c.h_bits, c.pushed, c.f[0], c.d_cc);
    HIP_TRY(h, hipGetLastError());
    // One launch per layer.  The layer count is only known on the device, so layers are enqueued in batches and
    // the frontier size of the next layer is read back after each batch (empty layers are no-ops).
    uint32_t level = 0;
    for (;;) {
        for (int b = 0; b < kCostBatch; ++b, ++level)
            hipLaunchKernelGGL(k_cost_level, dim3(kCostBlocks), dim3(kCostThreads), 0, s, V, R, c.ring_n, level, c.h_bits,
                               c.pushed, c.state, c.f[level & 1u], c.f[(level + 1u) & 1u], c.ring, c.d_cc);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(c.h_cc, c.d_cc, sizeof(CostCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        if (c.h_cc->frontier[level % 3u] == 0u) break;
        if (level > (1u << 24)) { h->err = "cost flood did not terminate"; return GNDT_ERR_HIP; }
    }
    if (c.h_cc->range_error) {
        h->err = "cost map: column indices beyond 32767 (mortonToXY decodes no further, Stopwatch.h:171-189)";
        return GNDT_ERR_KEY_RANGE;
    }
    if (c.h_cc->ring_overflow) {
        h->err = "cost map: a collision ring holds more than " + std::to_string(kRingCap) + " slopes (robot radius too large for this grid)";
        return GNDT_ERR_CAPACITY;
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
        st->goal_status = cc->goal_status; st->ring = (uint32_t)h->cost.ring_n; st->levels = cc->levels; st->reserved = 0;
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

// ---------------------------------------------------------------------------------------------
// one global map from a sharded cloud: shard -> statistics (PARTITION pipeline, no node table), and
// merged statistics (sorted by key) -> map
// ---------------------------------------------------------------------------------------------
int gndt_shard_stats_device(gndt_handle* h, const void* xyz_dev, size_t n, size_t stride_bytes, uint64_t first_idx_base,
                            gndt_stats* out, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!out || (!xyz_dev && n)) { h->err = "null argument"; return GNDT_ERR_INVALID; }
    if (first_idx_base + n >= 0x7FFFFFFFull) { h->err = "point index exceeds 31 bits"; return GNDT_ERR_INVALID; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->pending.active = false;
    h->last_stream = s;
    next_event_set(h);
    rc = -1;
    if (n >= (1u << 12) && h->P.strategy != GNDT_STRATEGY_ATOMIC && stride_bytes != 0) {
        auto& q = h->part;
        if ((rc = ensure_words(h, (n + 31) / 32 + 1))) return rc;
        if ((rc = ensure_part_counters(h))) return rc;
        auto& P = h->pending;
        P = gndt_handle::Pending{};
        P.xyz = xyz_dev; P.n = n; P.stride = stride_bytes; P.s = s;
        P.stats_only = true; P.first_base = (uint32_t)first_idx_base;
        P.nodes_est = h->P.max_nodes_hint ? h->P.max_nodes_hint : (q.nodes_learned ? q.nodes_learned : std::max<uint64_t>(n / 4, 1024));
        P.stage_want = std::max<uint64_t>(h->st_cap, std::max<uint64_t>(4096, n / 4));
        h->results_valid = false;
        const int prev = h->last_strategy;
        h->last_strategy = GNDT_STRATEGY_PARTITION;
        rc = partition_launch(h, P);
        if (rc == GNDT_OK) { P.active = true; rc = partition_resolve(h); }
        if (rc != -1) {
            if (rc) { h->last_strategy = prev; return rc; }
            if (h->h_cnt->err_key_range) {
                h->err = std::to_string(h->h_cnt->err_key_range) + " point(s) outside the key range";
                return GNDT_ERR_KEY_RANGE;
            }
            out->num_nodes = h->h_cnt->num_nodes;
            out->key = h->st_key; out->sums = h->st_sums; out->count = h->st_count; out->first_idx = h->st_first;
            return GNDT_OK;
        }
        h->last_strategy = prev;
    }
    // small shards, strategy ATOMIC, or too many nodes per bucket: through the node table
    h->pending.active = false;
    if ((rc = gndt_reset(h, hip_stream))) return rc;
    if ((rc = gndt_accumulate_device(h, xyz_dev, n, stride_bytes, first_idx_base, hip_stream))) return rc;
    return gndt_stats_export_device(h, out, hip_stream);
}

int gndt_finalize_stats_device(gndt_handle* h, const gndt_stats* in, uint64_t total_points, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!in || (in->num_nodes && (!in->key || !in->sums || !in->count || !in->first_idx))) { h->err = "null statistics"; return GNDT_ERR_INVALID; }
    if (in->num_nodes >= 0xFFFFFFFFull || total_points >= 0xFFFFFFFFull) { h->err = "too many nodes / points"; return GNDT_ERR_INVALID; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->pending.active = false;
    h->last_stream = s;
    next_event_set(h);
    auto& q = h->part;
    const uint64_t n = in->num_nodes;
    const uint64_t words = (std::max<uint64_t>(total_points, 64) + 31) / 32 + 1;
    if ((rc = ensure_part_counters(h))) return rc;
    if ((rc = ensure_stage(h, std::max<uint64_t>(1024, n)))) return rc;
    if ((rc = ensure_out(h, q.stage_cap))) return rc;
    if ((rc = ensure_words(h, words))) return rc;
    if (h->table_dirty) { if ((rc = do_reset(h, s))) return rc; }
    h->results_valid = false;
    mark(h, 0, s);
    hipLaunchKernelGGL(k_part_clear, dim3(grid_for(words, 256, 512)), dim3(256), 0, s, h->d_cnt, q.d_pc, q.bitmap, q.word_weight, (uint64_t)words,
                       (uint32_t*)nullptr, 0u);
    HIP_TRY(h, hipGetLastError());
    mark(h, 4, s);
    hipLaunchKernelGGL(k_stats_rows, dim3(grid_for(std::max<uint64_t>(n, 1))), dim3(kBlock), 0, s, in->key, in->sums, in->count,
                       in->first_idx, (uint32_t)n, grid_params(h), q.stage, q.ord_cf, q.ord_idx, ColumnOrder{q.bitmap, q.word_weight, q.ncol_at},
                       (uint64_t)words, h->d_cnt, q.d_pc);
    HIP_TRY(h, hipGetLastError());
    mark(h, 5, s);
    if ((rc = launch_order_and_emit(h, words, 5, s))) return rc;
    HIP_TRY(h, hipMemcpyAsync(q.h_pc, q.d_pc, sizeof(PartCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(h->h_cnt, h->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    h->results_valid = true;
    ++h->result_serial;
    h->map_in_table = false;
    h->last_strategy = GNDT_STRATEGY_PARTITION;
    h->stream_pos = total_points;
    return GNDT_OK;
}

// ---------------------------------------------------------------------------------------------
// input side: raw records -> packed xyz (gndt_pack.hpp)
// ---------------------------------------------------------------------------------------------
int gndt_pack_points_device(gndt_handle* h, const void* raw_dev, size_t n, const gndt_point_layout* layout,
                            float* xyz_out_dev, uint64_t* n_valid, void* hip_stream) {
    if (!h || !layout || !n_valid) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    *n_valid = 0;
    if (n == 0) return GNDT_OK;
    if (!raw_dev || !xyz_out_dev) { h->err = "null buffer"; return GNDT_ERR_INVALID; }
    if (n >= 0xFFFFFFFFull) { h->err = "too many points"; return GNDT_ERR_INVALID; }
    const PointLayout L{layout->point_step, layout->offset_x, layout->offset_y, layout->offset_z};
    if (L.point_step < 12 || (L.point_step & 3u) || ((L.off_x | L.off_y | L.off_z) & 3u) ||
        std::max(L.off_x, std::max(L.off_y, L.off_z)) + 4 > L.point_step) {
        h->err = "point layout: 4-byte aligned float fields inside a record of point_step bytes expected";
        return GNDT_ERR_INVALID;
    }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    int rc = partition_resolve(h);              // the buffers below are shared with a pending build's ordering pass
    if (rc) return rc;
    auto& q = h->part;
    const uint64_t words = ((n + 63) / 64) * 2;
    if ((rc = ensure_words(h, words))) return rc;
    h->incr_ok = false;                        // the order's bitmap is used as scratch here
    if (!h->d_nvalid) {
        HIP_TRY(h, hipMalloc(&h->d_nvalid, sizeof(uint32_t)));
        HIP_TRY(h, hipHostMalloc(&h->h_nvalid, sizeof(uint32_t)));
    }
    const unsigned char* raw = static_cast<const unsigned char*>(raw_dev);
    hipLaunchKernelGGL(k_pack_flags, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, raw, (uint64_t)n, L,
                       reinterpret_cast<unsigned long long*>(q.bitmap));
    const uint32_t nbw = (uint32_t)((words + kScanChunk - 1) / kScanChunk);
    hipLaunchKernelGGL(k_scan_reduce<true>, dim3(nbw), dim3(kScanThreads), 0, s, q.bitmap, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words);
    hipLaunchKernelGGL(k_scan_apply<true>, dim3(nbw), dim3(kScanThreads), 0, s, q.bitmap, (const uint32_t*)nullptr,
                       (uint32_t)words, q.bsum_words, q.word_base);
    hipLaunchKernelGGL(k_pack_write, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, raw, (uint64_t)n, L, q.bitmap,
                       q.word_base, xyz_out_dev, h->d_nvalid);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_nvalid, h->d_nvalid, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    *n_valid = *h->h_nvalid;
    return GNDT_OK;
}

int gndt_build_cloud(gndt_handle* h, const void* raw_host, size_t n, const gndt_point_layout* layout) {
    if (!h || !layout) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!raw_host && n) { h->err = "null input"; return GNDT_ERR_INVALID; }
    if (n == 0) { h->err = "empty cloud: there is no first point to take the origin from (receiver.cpp:145)"; return GNDT_ERR_INVALID; }
    int rc = stage_host_input(h, raw_host, n, layout->point_step, h->own_stream);
    if (rc) return rc;
    if (n > h->packed_cap) {
        if (h->packed) (void)hipFree(h->packed);
        h->packed = nullptr; h->packed_cap = 0;
        HIP_TRY(h, hipMalloc(&h->packed, n * 12));
        h->packed_cap = n;
    }
    uint64_t valid = 0;
    rc = gndt_pack_points_device(h, h->stage, n, layout, h->packed, &valid, h->own_stream);
    if (rc) return rc;
    if (valid == 0) { h->err = "no finite point in the cloud"; return GNDT_ERR_INVALID; }
    float origin[3];
    HIP_TRY(h, hipMemcpy(origin, h->packed, sizeof origin, hipMemcpyDeviceToHost));
    if (h->table_dirty) { rc = gndt_reset(h, h->own_stream); if (rc) return rc; }
    rc = gndt_set_origin(h, origin);
    if (rc) return rc;
    rc = gndt_build_device(h, h->packed + 3, (size_t)valid - 1, 12, h->own_stream);
    if (rc) return rc;
    return gndt_sync(h, nullptr, nullptr, nullptr);
}

int gndt_stats_export_device(gndt_handle* h, gndt_stats* out, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!out) return GNDT_ERR_INVALID;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    if (!h->map_in_table) {
        h->err = "the current map was built by the PARTITION strategy, which keeps no additive state (use strategy ATOMIC)";
        return GNDT_ERR_INVALID;
    }
    if (h->cap == 0) { rc = alloc_table(h, cap_for_nodes(1024), s); if (rc) return rc; }
    rc = fetch_counters(h, s);
    if (rc) return rc;
    if (h->h_cnt->err_table_full) { h->err = "node table full: raise gndt_params.max_nodes_hint"; return GNDT_ERR_CAPACITY; }
    const uint32_t C = h->h_cnt->num_nodes;
    rc = ensure_stats_buffers(h, C);
    if (rc) return rc;
    if (C) {
        hipLaunchKernelGGL(k_stats_export, dim3(grid_for(C)), dim3(kBlock), 0, s, h->keys, h->acc, h->node_slot, h->d_cnt,
                           h->st_key, h->st_sums, h->st_count, h->st_first);
        HIP_TRY(h, hipGetLastError());
    }
    out->num_nodes = C;
    out->key = h->st_key; out->sums = h->st_sums; out->count = h->st_count; out->first_idx = h->st_first;
    return GNDT_OK;
}

int gndt_stats_merge_device(gndt_handle* h, const gndt_stats* in, void* hip_stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!in) return GNDT_ERR_INVALID;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    h->last_stream = s;
    if (!h->map_in_table) {
        h->err = "the current map was built by a PARTITION strategy and does not live in the node table: call gndt_reset first";
        return GNDT_ERR_INVALID;
    }
    h->pending.active = false;
    rc = ensure_capacity_for(h, in->num_nodes, s);
    if (rc) return rc;
    if (in->num_nodes) {
        hipLaunchKernelGGL(k_stats_merge, dim3(grid_for(in->num_nodes)), dim3(kBlock), 0, s, h->keys, h->acc, h->cap - 1,
                           h->node_slot, h->index_of_slot, in->key, in->sums, in->count, in->first_idx, (uint64_t)in->num_nodes,
                           h->d_cnt);
        h->incr_ok = false;
        HIP_TRY(h, hipGetLastError());
        h->table_dirty = true;
        h->results_valid = false;
        // the merged first indices tell how far the point stream reaches (sizes the column-order bitmap);
        // the exchange path may wait for the host, and learns the exact node count on the way
        rc = fetch_counters(h, s);
        if (rc) return rc;
        h->stream_pos = std::max<uint64_t>(h->stream_pos, h->h_cnt->stream_pos);
        h->nodes_bound = h->h_cnt->num_nodes;
    }
    return GNDT_OK;
}

int gndt_set_profiling(gndt_handle* h, int enable) {
    if (!h) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (enable)
        for (auto& set : h->ev)
            for (auto& e : set)
                if (!e) HIP_TRY(h, hipEventCreate(&e));
    for (auto& set : h->ev_recorded)
        for (auto& r : set) r = false;
    h->prof = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return GNDT_OK;
}

// Mean duration of every phase over the builds recorded since the last call (at most the last kEvSets).
int gndt_get_phase_times(gndt_handle* h, double ms_out[GNDT_NUM_PHASES]) {
    if (!h || !ms_out) return GNDT_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    for (int i = 0; i < GNDT_NUM_PHASES; ++i) {
        double sum = 0.0;
        int cnt = 0;
        for (int k = 0; k < gndt_handle::kEvSets; ++k) {
            if (!(h->ev_recorded[k][i] && h->ev_recorded[k][i + 1])) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[k][i], h->ev[k][i + 1]) == hipSuccess) { sum += ms; ++cnt; }
        }
        ms_out[i] = cnt ? sum / cnt : -1.0;
    }
    for (auto& set : h->ev_recorded)
        for (auto& r : set) r = false;
    return GNDT_OK;
}

int gndt_debug_bucket_phases(gndt_handle* h, double cycles_out[10], uint32_t* buckets_out) {
    if (!h || !cycles_out) return GNDT_ERR_INVALID;
    auto& q = h->part;
    if (!q.dbg || !q.last_buckets) { h->err = "no stamps: set GNDT_STAMPS=1 before a PARTITION build"; return GNDT_ERR_INVALID; }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    std::vector<unsigned long long> t((size_t)q.last_buckets * 16);
    HIP_TRY(h, hipMemcpy(t.data(), q.dbg, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 10; ++k) cycles_out[k] = 0.0;
    for (uint32_t b = 0; b < q.last_buckets; ++b) {
        const unsigned long long* s = &t[(size_t)b * 16];
        for (int k = 0; k < 6; ++k) cycles_out[k] += (double)(s[k + 1] - s[k]);
        // sub-phases of the accumulate phase, summed over the bucket's chunks: load wait, classify, scan+scatter, reduce
        for (int k = 0; k < 4; ++k) cycles_out[6 + k] += (double)s[8 + k];
    }
    for (int k = 0; k < 10; ++k) cycles_out[k] /= q.last_buckets;
    if (buckets_out) *buckets_out = q.last_buckets;
    return GNDT_OK;
}

int gndt_last_strategy(const gndt_handle* h) { return h ? h->last_strategy : GNDT_STRATEGY_AUTO; }

int gndt_device_info(int32_t device_id, char name_out[128], int32_t* compute_units, uint64_t* hbm_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return GNDT_ERR_NO_DEVICE;
    if (name_out) {
        name_out[0] = 0;
        (void)hipDeviceGetName(name_out, 128, device_id);
    }
    if (compute_units) {
        int cu = 0;
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id);
        *compute_units = cu;
    }
    if (hbm_bytes) {
        size_t total = 0;
        (void)hipDeviceTotalMem(&total, device_id);
        *hbm_bytes = total;
    }
    return GNDT_OK;
}

}  // extern "C"
