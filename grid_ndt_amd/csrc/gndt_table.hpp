// gndt_table.hpp — finalisation of the persistent node table (strategy ATOMIC): from the additive statistics to
// result rows in reference order, without ever waiting for the host.
//
//   k_tab_begin     zero the per-finalize counters and the column-first bitmap; forget the column lists of the
//                   nodes that were finalised before (their columns are rebuilt below)
//   k_tab_columns   per node: fp32 mean-z, column registration (first-seen = min over its nodes), linked list
//   k_tab_rows      per node: slope label (OcNode::isSlope, map2D.h:66-108) and index in column by walking the
//                   column's list; mean + fp64 scatter -> 128-B staging row; bit per column-first index
//   then the partition path's ordering (k_scan_*, k_order_*) and k_emit_rows (gndt_partition.hpp)
//   (k_emit_rows also does the end-of-frame bookkeeping: how many nodes now own a column entry, next epoch, stream position)
//
// Incremental updates (gndt_update_device; SURVEY §8(f) rank 2: "per-frame re-labelling of touched columns"): the staging
// rows, the per-node order keys and the column order (bitmap, word weights) persist between finalisations, k_accumulate
// lists the nodes a frame touched, and only their columns are redone:
//   k_tab_touch          touched nodes: fp32 mean-z; new nodes join their column; the columns go on a list
//   k_tab_expand         the nodes of the listed columns as a flat work list
//   k_tab_rows_touched   every listed node: label, index in column, moments -> its staging row; the column's weight in
//                        the order changes by the nodes it gained
// followed by the same ordering + emit over the whole map (rows move when a column in front of them grows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_partition.hpp"

namespace gndt {

// What a node's look at its column needs of the OTHER nodes, by node index in one 16-byte load: a walk over a column's list
// then costs one dependent load per step (the next pointer) with this record beside it, instead of node -> slot -> {first-seen
// index, key, mean z} two levels deep (the touched-column pass of a 131 k-point frame: 40 -> ~25 us).  Written whenever the
// node's SlotAux is (k_tab_columns, k_tab_touch); first-seen index and z level never change once the node exists.
struct alignas(16) NodeInfo {
    uint32_t first; int32_t sz; float mean_z; uint32_t flags;
};

struct TableView {
    uint64_t* keys; NodeAcc* acc; SlotAux* aux; NodeInfo* ninfo;
    uint64_t* col_keys; uint32_t* col_first; uint32_t* col_cnt; uint32_t* col_head;
    uint32_t* node_slot; uint32_t* col_slot_of_node; uint32_t* node_next;
    uint32_t* index_of_slot; uint32_t* touch_epoch; uint32_t* col_epoch;      // incremental updates
    uint32_t* touched; uint32_t* touched_cols;
    uint32_t cap_mask;
};

// the column-table slot of a column key, inserting it if new
__device__ __forceinline__ uint32_t tab_column_slot(const TableView& T, uint64_t ck) {
    uint32_t cs = (uint32_t)mix64(ck) & T.cap_mask;
    for (uint32_t probe = 0; probe <= T.cap_mask; ++probe) {
        const uint64_t k = T.col_keys[cs];
        if (k == ck) break;
        if (k == kEmptyKey) {
            const unsigned long long old = atomicCAS((unsigned long long*)&T.col_keys[cs], (unsigned long long)kEmptyKey,
                                                     (unsigned long long)ck);
            if (old == kEmptyKey || old == ck) break;
        }
        cs = (cs + 1) & T.cap_mask;
    }
    return cs;
}

// The staging row of node i from the table's current state: slope label (OcNode::isSlope, map2D.h:66-108) and index in
// column by walking the column's list, mean + fp64 scatter from the additive statistics.
__device__ __forceinline__ void tab_make_row(const TableView& T, const GridParams& P, uint32_t i, StageRow& row) {
    const uint32_t s = T.node_slot[i];
    const uint64_t key = T.keys[s];
    const NodeAcc a = T.acc[s];
    const uint32_t cs = T.col_slot_of_node[i];
    uint32_t fl = T.aux[s].flags & 1u;
    const float cz = T.aux[s].mean_z;
    unpack_key(key, row.sx, row.sy, row.sz);
    const int za = level_above(row.sz), zb = level_below(row.sz);
    uint32_t icol = 0;
    bool up = false, down = false;
    for (uint32_t t = T.col_head[cs]; t != 0xFFFFFFFFu; t = T.node_next[t]) {
        if (t == i) continue;
        const NodeInfo tx = T.ninfo[t];
        const uint32_t tf = tx.first;
        icol += (tf < a.first) ? 1u : 0u;
        const int tz = tx.sz;
        if (tz == za || tz == zb) {
            const bool visited = tf < a.first && (tx.flags & 1u);
            const float oz = visited ? tx.mean_z : 0.f;
            const bool far = fabsf(oz - cz) > P.slope_interval;
            if (tz == za) up = up || far; else down = down || far;
        }
    }
    if (fl & 1u) {
        bool slope = true;
        if (P.demand == 0) slope = !up; else down = false;
        if (slope) { fl |= 2u; if (down) fl |= 4u; }
    }
    row.count = a.count; row.first = a.first; row.flags = fl;
    for (int k = 0; k < 3; ++k) row.mean[k] = 0.f;
    for (int k = 0; k < 6; ++k) row.scatter[k] = 0.0;
    if (fl & 1u) {
        const double c[3] = {axis_centre(row.sx, P.ox, P.grid_len), axis_centre(row.sy, P.oy, P.grid_len),
                             axis_centre(row.sz, P.oz, P.z_len)};
        node_moments(a.count, a.s, c, row.mean, row.scatter);
    }
    row.col_first = T.col_first[cs]; row.idx_in_col = icol; row.ncol = T.col_cnt[cs];
}

static __global__ void __launch_bounds__(kBlock) k_tab_begin(TableView T, Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                      uint32_t* __restrict__ bitmap, uint32_t* __restrict__ word_weight,
                                                      uint64_t words, uint32_t raise_to) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (uint64_t)gridDim.x * blockDim.x;
    if (gid == 0 && raise_to) cnt->stream_pos = max(cnt->stream_pos, raise_to);
    if (gid == 0) { cnt->num_columns = 0; cnt->num_slopes = 0; pc->lds_overflow = 0; pc->stage_overflow = 0; pc->index_overflow = 0; pc->part_overflow = 0; pc->small_fallback = 0; cnt->first_word = 0u; cnt->n_work = 0u; }
    for (uint64_t i = gid; i < words; i += gsz) { bitmap[i] = 0u; word_weight[i] = 0u; }
    const uint32_t np = cnt->prev_nodes;
    for (uint64_t i = gid; i < np; i += gsz) {
        const uint32_t cs = T.col_slot_of_node[i];
        T.col_head[cs] = 0xFFFFFFFFu;
        T.col_cnt[cs] = 0u;
    }
}

static __global__ void __launch_bounds__(kBlock) k_tab_columns(TableView T, GridParams P, const Counters* __restrict__ cnt) {
    const uint32_t n = cnt->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = T.node_slot[i];
        const uint64_t key = T.keys[s];
        const NodeAcc a = T.acc[s];
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        SlotAux x;
        x.flags = 0u; x.mean_z = 0.f;
        if (a.count >= (uint32_t)P.min_points) { x.mean_z = node_mean_z(a.count, a.s[2], axis_centre(sz, P.oz, P.z_len)); x.flags = 1u; }
        T.aux[s] = x;
        T.ninfo[i] = NodeInfo{a.first, sz, x.mean_z, x.flags};
        const uint32_t cs = tab_column_slot(T, column_key(key));
        T.col_slot_of_node[i] = cs;
        atomicMin(&T.col_first[cs], a.first);
        atomicAdd(&T.col_cnt[cs], 1u);
        T.node_next[i] = atomicExch(&T.col_head[cs], i);
    }
}

static __global__ void __launch_bounds__(kBlock) k_tab_rows(TableView T, GridParams P, StageRow* __restrict__ stage, uint32_t stage_cap,
                                                     uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                     ColumnOrder O, uint64_t words, Counters* __restrict__ cnt,
                                                     PartCounters* __restrict__ pc) {
    __shared__ uint32_t s_slopes, s_cols;
    if (threadIdx.x == 0) { s_slopes = 0; s_cols = 0; }
    __syncthreads();
    const uint32_t n = cnt->num_nodes;
    if (n > stage_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pc->stage_overflow, n); return; }
    uint32_t my_slopes = 0, my_cols = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        StageRow row;
        tab_make_row(T, P, i, row);
        if (row.flags & 2u) ++my_slopes;
        stage[i] = row;
        ord_cf[i] = row.col_first;
        ord_idx[i] = row.idx_in_col;
        if (row.idx_in_col == 0) {
            // the bitmap was sized from the host's view of the stream (max_points_hint for captured updates)
            if ((uint64_t)(row.col_first >> 5) < words) note_column(O, row.col_first, row.ncol);
            else atomicAdd(&pc->index_overflow, 1u);
            ++my_cols;
        }
    }
    // one memory-side atomic per block and counter
    if (my_slopes) atomicAdd(&s_slopes, my_slopes);
    if (my_cols) atomicAdd(&s_cols, my_cols);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_slopes) atomicAdd(&cnt->num_slopes, s_slopes);
        if (s_cols) atomicAdd(&cnt->num_columns, s_cols);
    }
}


// ---------------------------------------------------------------------------------------------
// Small maps: the whole finalisation in ONE workgroup (round 4).
// A depth-camera frame at the launch cells — what BASELINE configs[0]'s .pcd files are — has a few hundred nodes; taking them
// through k_tab_begin / k_tab_columns / k_tab_rows / the scan / the destination pass / k_emit_rows costs six launches of 4-20 us
// each for work that fits one workgroup's LDS.  Here thread i IS node i (at most kSmallMapNodes of them): columns are found in an
// LDS table and sorted by their first-seen index (a bitonic sort of at most 1024 keys), the node counts in that order are scanned
// into first rows, a column's nodes are put next to each other, and every node walks its own column once for its index in the
// column and its two z neighbours — then writes its result row where it belongs.  (A first version compared every node with
// every other node of the map: 562^2 pairs on ONE compute unit took 0.1 ms, slower than the six launches it replaced.)  No staging rows, no order arrays, no bitmap: the incremental finalisation of gndt_update* cannot
// continue from such a map and takes its full path (the host clears gndt_handle::incr_ok).
// More nodes than the workgroup has threads: PartCounters::small_fallback is raised, nothing is written, the host runs the
// regular kernels (an eager build does so at once; a replayed hipGraph reports GNDT_ERR_CAPACITY like any other overflow).
// ---------------------------------------------------------------------------------------------
constexpr int kSmallMapNodes = 1024;
constexpr int kSmallColSlots = 2048;
struct SmallMapLds {
    uint4 colnodes[kSmallMapNodes];             // the nodes of every column next to each other: {first-seen index, z level, fp32 mean z, -}
    unsigned long long ckey[kSmallColSlots];    // column table: key (kEmptyKey = free)
    uint32_t ccf[kSmallColSlots];               //   first-seen index of the column (min over its nodes); later: nodes placed so far
    uint32_t ccnt[kSmallColSlots];              //   nodes of the column; later: (first row << 16) | nodes
    unsigned long long sortk[kSmallMapNodes];   // the occupied columns as (first-seen index << 16 | slot), sorted: position = rank
    uint32_t by_rank[kSmallMapNodes];           // node count of the column of rank r -> exclusive prefix = its first row
    uint32_t wsum[kSmallMapNodes / 64];
    uint32_t n_cols, n_slopes;
};

static __global__ void __launch_bounds__(kSmallMapNodes) k_small_finalize(TableView T, GridParams P, OutView out, uint32_t* __restrict__ row_ncol,
                                                                   Counters* cnt, PartCounters* __restrict__ pc, Counters* __restrict__ host_cnt,
                                                                   PartCounters* __restrict__ host_pc, uint32_t raise_to, uint32_t out_cap,
                                                                   uint32_t capture_id) {
    __shared__ SmallMapLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = cnt->num_nodes;
    if (n > (uint32_t)kSmallMapNodes || n > out_cap) {          // uniform: not a small map after all
        if (tid == 0) {
            pc->small_fallback = 1u;
            if (host_cnt) *host_cnt = *cnt;
            if (host_pc) { *host_pc = *pc; host_pc->capture_id = capture_id; }
        }
        return;
    }
    for (int c = tid; c < kSmallColSlots; c += kSmallMapNodes) { L.ckey[c] = kEmptyKey; L.ccf[c] = 0xFFFFFFFFu; L.ccnt[c] = 0u; }
    L.sortk[tid] = ~0ull;
    if (tid == 0) { L.n_cols = 0; L.n_slopes = 0; }
    __syncthreads();
    // ---- the node, its column ----
    const bool live = (uint32_t)tid < n;
    uint64_t key = 0;
    NodeAcc a;
    for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
    a.count = 0; a.first = 0xFFFFFFFFu;
    int sx = 0, sy = 0, sz = 0;
    float cz = 0.f;
    uint32_t col = 0, fl = 0;
    if (live) {
        const uint32_t slot = T.node_slot[tid];
        key = T.keys[slot];
        a = T.acc[slot];
        unpack_key(key, sx, sy, sz);
        if (a.count >= (uint32_t)P.min_points) { cz = node_mean_z(a.count, a.s[2], axis_centre(sz, P.oz, P.z_len)); fl = 1u; }
        const unsigned long long ck = column_key(key);
        col = (uint32_t)mix64(ck) & (uint32_t)(kSmallColSlots - 1);
        for (int probe = 0; probe < kSmallColSlots; ++probe) {      // (terminates: twice as many slots as nodes)
            const unsigned long long k = L.ckey[col];
            if (k == ck) break;
            if (k == kEmptyKey) {
                const unsigned long long old = atomicCAS(&L.ckey[col], (unsigned long long)kEmptyKey, ck);
                if (old == kEmptyKey || old == ck) break;
            }
            col = (col + 1) & (uint32_t)(kSmallColSlots - 1);
        }
        atomicMin(&L.ccf[col], a.first);
        atomicAdd(&L.ccnt[col], 1u);
    }
    __syncthreads();
    // ---- the occupied columns, compacted as sort keys (thread = column slot, two passes over the table) ----
    for (int c = tid; c < kSmallColSlots; c += kSmallMapNodes) {
        const bool occ = L.ccnt[c] != 0u;
        const unsigned long long m = __ballot(occ);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&L.n_cols, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (occ) L.sortk[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)L.ccf[c] << 16) | (unsigned long long)c;
    }
    __syncthreads();
    const uint32_t K = L.n_cols;
    // ---- columns in first-seen order (first-seen indices of distinct columns are distinct points).  Few columns: every column
    //      counts the columns seen before it (K^2 / K threads, all reading the same word at a time: LDS broadcasts) and moves to
    //      that place; many: a bitonic sort of the keys, padded to the next power of two (the padding sorts to the end) ----
    if (K <= 256u) {
        unsigned long long mine = ~0ull;
        uint32_t rank = 0;
        if ((uint32_t)tid < K) {
            mine = L.sortk[tid];
            for (uint32_t j = 0; j < K; ++j) rank += (L.sortk[j] < mine) ? 1u : 0u;
        }
        __syncthreads();
        if ((uint32_t)tid < K) L.sortk[rank] = mine;
        __syncthreads();
    } else {
        uint32_t N2 = 512;
        while (N2 < K) N2 <<= 1;
        for (uint32_t k = 2; k <= N2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const uint32_t ixj = (uint32_t)tid ^ j;
                if (ixj > (uint32_t)tid && ixj < N2) {
                    const unsigned long long x = L.sortk[tid], y = L.sortk[ixj];
                    const bool up = ((uint32_t)tid & k) == 0;
                    if ((x > y) == up) { L.sortk[tid] = y; L.sortk[ixj] = x; }
                }
                __syncthreads();
            }
        }
    }
    // ---- the node counts in rank order, scanned: the first row of every column ----
    uint32_t my_cslot = 0, my_ccnt = 0;
    if ((uint32_t)tid < K) { my_cslot = (uint32_t)(L.sortk[tid] & 0xFFFFull); my_ccnt = L.ccnt[my_cslot]; }
    {
        const uint32_t v = my_ccnt;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) L.wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - v;
        for (int w = 0; w < wave; ++w) base += L.wsum[w];
        if ((uint32_t)tid < K) { L.ccnt[my_cslot] = (base << 16) | my_ccnt; L.ccf[my_cslot] = 0u; }      // first row | nodes (both fit 16 bits); nodes placed: 0
    }
    __syncthreads();
    // ---- a column's nodes next to each other (the order inside the array is free), then every node looks at its own column ----
    uint32_t cinfo = 0;
    if (live) {
        cinfo = L.ccnt[col];
        const uint32_t k = atomicAdd(&L.ccf[col], 1u);
        L.colnodes[(cinfo >> 16) + k] = make_uint4(a.first, (uint32_t)sz, __float_as_uint(cz), 0u);
    }
    __syncthreads();
    if (live) {
        const int za = level_above(sz), zb = level_below(sz);
        const uint32_t cbase = cinfo >> 16, ncol = cinfo & 0xFFFFu;
        uint32_t icol = 0;
        bool up = false, down = false;
        for (uint32_t j = 0; j < ncol; ++j) {
            const uint4 o = L.colnodes[cbase + j];       // {first-seen, z level, mean z}
            if (o.x == a.first) continue;                // (this node itself: first-seen indices of distinct nodes are distinct points)
            icol += (o.x < a.first) ? 1u : 0u;
            const int tz = (int)o.y;
            if (tz == za || tz == zb) {
                const float oz = (o.x < a.first) ? __uint_as_float(o.z) : 0.f;        // "visited": seen earlier AND has statistics
                const bool far = fabsf(oz - cz) > P.slope_interval;
                if (tz == za) up = up || far; else down = down || far;
            }
        }
        if (fl & 1u) {
            bool slope = true;
            if (P.demand == 0) slope = !up; else down = false;
            if (slope) { fl |= 2u; if (down) fl |= 4u; atomicAdd(&L.n_slopes, 1u); }
        }
        const uint32_t r = cbase + icol;
        out.sx[r] = sx; out.sy[r] = sy; out.sz[r] = sz;
        out.count[r] = a.count; out.first_idx[r] = a.first; out.flags[r] = fl;
        row_ncol[r] = icol == 0u ? ncol : 0u;
        float mean[3] = {0.f, 0.f, 0.f}, rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
        double S[6] = {0, 0, 0, 0, 0, 0};
        if (fl & 1u) {
            const double c[3] = {axis_centre(sx, P.ox, P.grid_len), axis_centre(sy, P.oy, P.grid_len), axis_centre(sz, P.oz, P.z_len)};
            node_moments(a.count, a.s, c, mean, S);
            node_rough_normal(S, rough, normal);
        }
        out.rough[r] = rough;
        for (int k = 0; k < 3; ++k) { out.mean[3 * r + k] = mean[k]; out.normal[3 * r + k] = normal[k]; }
        for (int k = 0; k < 6; ++k) out.cov[6 * r + k] = (float)S[k];
    }
    __syncthreads();
    if (tid == 0) {
        cnt->num_columns = K; cnt->num_slopes = L.n_slopes;
        if (raise_to) cnt->stream_pos = max(cnt->stream_pos, raise_to);
        cnt->n_touched = 0; cnt->n_tcols = 0; cnt->n_work = 0; cnt->first_word = 0u;
        cnt->epoch = cnt->epoch + 1u;                      // (prev_nodes stays 0: no node owns an entry of the HBM column table)
        pc->lds_overflow = 0; pc->stage_overflow = 0; pc->index_overflow = 0; pc->part_overflow = 0; pc->small_fallback = 0;
        if (host_cnt) *host_cnt = *cnt;
        if (host_pc) { *host_pc = *pc; host_pc->capture_id = capture_id; }
    }
}

// The end-of-frame bookkeeping k_emit_rows does for the table path, as a launch of its own: what a frame of a handle in
// deferred-emit mode (gndt_set_deferred_emit) ends with instead of the ordering and the emit pass.
static __global__ void k_tab_end(Counters* cnt, const PartCounters* __restrict__ pc, Counters* __restrict__ host_cnt, PartCounters* __restrict__ host_pc,
                          uint32_t advance, uint32_t capture_id) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    cnt->prev_nodes = cnt->num_nodes;
    cnt->n_touched = 0; cnt->n_tcols = 0;
    cnt->epoch = cnt->epoch + 1u;
    cnt->stream_pos += advance;
    if (host_cnt) *host_cnt = *cnt;
    if (host_pc) { *host_pc = *pc; host_pc->capture_id = capture_id; }
}

// Room for `mine` entries of this lane in a list whose length is *counter: ONE counter atomic per block (a counter is one
// word; same-address atomics retire at ~90 per microsecond at the memory side, so one per wave is felt in kernels this
// short).  Every thread of the block calls it, the same number of times.  s: kBlock / 64 + 1 words of LDS.
__device__ __forceinline__ uint32_t block_list_reserve(uint32_t mine, uint32_t* __restrict__ counter, uint32_t* s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < kBlock / 64; ++w) total += s[w];
        s[kBlock / 64] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s[kBlock / 64] + incl - mine;
    for (int w = 0; w < wave; ++w) base += s[w];
    __syncthreads();                                   // (s is written again by the next call)
    return base;
}

// Incremental finalisation, step 1: the nodes the frame touched.
static __global__ void __launch_bounds__(kBlock) k_tab_touch(TableView T, GridParams P, Counters* __restrict__ cnt,
                                                      PartCounters* __restrict__ pc) {
    const uint32_t n = cnt->n_touched, np = cnt->prev_nodes, epoch = cnt->epoch;
    // (stage_overflow and index_overflow stay set once raised: an incremental finalisation builds on the rows and the
    //  column order of the previous ones, so a frame that could not be recorded invalidates the map until a reset)
    if (blockIdx.x == 0 && threadIdx.x == 0) { pc->lds_overflow = 0; pc->part_overflow = 0; cnt->first_word = 0xFFFFFFFFu; cnt->n_work = 0u; }   // (both are next used by later kernels)
    __shared__ uint32_t s_res[kBlock / 64 + 1];
    const uint32_t n_round = (n + (uint32_t)kBlock - 1u) / (uint32_t)kBlock * (uint32_t)kBlock;   // whole blocks for the aggregated list append
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_round; j += gridDim.x * blockDim.x) {
        bool first = false;
        uint32_t cs = 0;
        if (j < n) {
            const uint32_t s = T.touched[j];
            const uint32_t i = T.index_of_slot[s];
            const uint64_t key = T.keys[s];
            const NodeAcc a = T.acc[s];
            const int sz = (int)(key & 0x3FFFFFu) - (1 << 21);
            SlotAux x;
            x.flags = 0u; x.mean_z = 0.f;
            if (a.count >= (uint32_t)P.min_points) { x.mean_z = node_mean_z(a.count, a.s[2], axis_centre(sz, P.oz, P.z_len)); x.flags = 1u; }
            T.aux[s] = x;
            T.ninfo[i] = NodeInfo{a.first, sz, x.mean_z, x.flags};
            if (i >= np) {                                     // a node born in this frame joins its column
                cs = tab_column_slot(T, column_key(key));
                T.col_slot_of_node[i] = cs;
                atomicMin(&T.col_first[cs], a.first);
                atomicAdd(&T.col_cnt[cs], 1u);
                T.node_next[i] = atomicExch(&T.col_head[cs], i);
            } else {
                cs = T.col_slot_of_node[i];
            }
            first = atomicExch(&T.col_epoch[cs], epoch) != epoch;
        }
        const uint32_t at = block_list_reserve(first ? 1u : 0u, &cnt->n_tcols, s_res);
        if (first) T.touched_cols[at] = cs;
    }
}

// Incremental finalisation, step 2: the nodes of the touched columns, as a flat work list (it reuses touched[], which
// step 1 has consumed).  A block adds up its columns' node counts and reserves the space with one atomic.
static __global__ void __launch_bounds__(kBlock) k_tab_expand(TableView T, Counters* __restrict__ cnt) {
    __shared__ uint32_t s_res[kBlock / 64 + 1];
    const uint32_t nc = cnt->n_tcols;
    const uint32_t nc_round = (nc + (uint32_t)kBlock - 1u) / (uint32_t)kBlock * (uint32_t)kBlock;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc_round; c += gridDim.x * blockDim.x) {
        uint32_t cs = 0, n = 0;
        if (c < nc) { cs = T.touched_cols[c]; n = T.col_cnt[cs]; }
        uint32_t base = block_list_reserve(n, &cnt->n_work, s_res);
        if (c < nc)
            for (uint32_t t = T.col_head[cs]; t != 0xFFFFFFFFu; t = T.node_next[t]) T.touched[base++] = t;
    }
}

// Incremental finalisation, step 3: a fresh staging row for every listed node; the first node of a column also updates
// the column's place in the order by what the column gained.
static __global__ void __launch_bounds__(kBlock) k_tab_rows_touched(TableView T, GridParams P, StageRow* __restrict__ stage,
                                                             uint32_t stage_cap, uint32_t* __restrict__ ord_cf,
                                                             uint32_t* __restrict__ ord_idx, ColumnOrder O, uint64_t words,
                                                             Counters* __restrict__ cnt, PartCounters* __restrict__ pc) {
    __shared__ int s_slopes;
    __shared__ uint32_t s_cols, s_w0;
    if (threadIdx.x == 0) { s_slopes = 0; s_cols = 0; s_w0 = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t my_w0 = 0xFFFFFFFFu;                   // first bitmap word whose columns changed size (one memory-side atomic per block)
    const uint32_t nw = cnt->n_work, np = cnt->prev_nodes;
    if (cnt->num_nodes > stage_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&pc->stage_overflow, cnt->num_nodes); return; }
    int my_slopes = 0;
    uint32_t my_cols = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nw; j += gridDim.x * blockDim.x) {
        const uint32_t t = T.touched[j];
        const uint32_t old_flags = (t < np) ? stage[t].flags : 0u;
        StageRow row;
        tab_make_row(T, P, t, row);
        my_slopes += (int)((row.flags >> 1) & 1u) - (int)((old_flags >> 1) & 1u);
        stage[t] = row;
        ord_cf[t] = row.col_first;
        ord_idx[t] = row.idx_in_col;
        if (row.idx_in_col == 0) {
            // the column in the order: a stream only appends (first-seen indices grow), so cf is stable once the column exists
            const uint32_t cf = row.col_first, w = cf >> 5, bit = 1u << (cf & 31u);
            if ((uint64_t)w >= words) { atomicAdd(&pc->index_overflow, 1u); continue; }
            const bool had = (O.bitmap[w] & bit) != 0u;
            const uint32_t old_n = had ? O.ncol_at[cf] : 0u;
            if (!had) { atomicOr(&O.bitmap[w], bit); ++my_cols; }
            if (row.ncol != old_n) { atomicAdd(&O.word_weight[w], row.ncol - old_n); my_w0 = min(my_w0, w); }   // rows from this word on move
            O.ncol_at[cf] = row.ncol;
        }
    }
    if (my_slopes) atomicAdd(&s_slopes, my_slopes);
    if (my_cols) atomicAdd(&s_cols, my_cols);
    if (my_w0 != 0xFFFFFFFFu) atomicMin(&s_w0, my_w0);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_slopes) atomicAdd(&cnt->num_slopes, (uint32_t)s_slopes);     // two's complement: also subtracts
        if (s_cols) atomicAdd(&cnt->num_columns, s_cols);
        if (s_w0 != 0xFFFFFFFFu) atomicMin(&cnt->first_word, s_w0);
    }
}

// ---------------------------------------------------------------------------------------------
// k_stats_rows: merged statistics of a global map (unique nodes SORTED BY KEY, so the nodes of a column are
// adjacent) -> staging rows, without any table: the column of node i is the run of equal column keys around
// it; label, index in column and the column's first-seen index come from one walk over that run.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(kBlock) k_stats_rows(const uint64_t* __restrict__ key, const double* __restrict__ sums,
                                                       const uint32_t* __restrict__ count, const uint32_t* __restrict__ first,
                                                       uint32_t n, GridParams P, StageRow* __restrict__ stage,
                                                       uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                       ColumnOrder O, uint64_t words, Counters* __restrict__ cnt,
                                                       PartCounters* __restrict__ pc) {
    __shared__ uint32_t s_slopes, s_cols;
    if (threadIdx.x == 0) { s_slopes = 0; s_cols = 0; }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->num_nodes = n;
    uint32_t my_slopes = 0, my_cols = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t k = key[i], ck = column_key(k);
        StageRow row;
        unpack_key(k, row.sx, row.sy, row.sz);
        const uint32_t my_first = first[i], my_count = count[i];
        const double cz_centre = axis_centre(row.sz, P.oz, P.z_len);
        uint32_t fl = 0;
        float cz = 0.f;
        if (my_count >= (uint32_t)P.min_points) { cz = node_mean_z(my_count, sums[9 * (size_t)i + 2], cz_centre); fl = 1u; }
        const int za = level_above(row.sz), zb = level_below(row.sz);
        uint32_t lo = i, hi = i + 1;
        while (lo > 0 && column_key(key[lo - 1]) == ck) --lo;
        while (hi < n && column_key(key[hi]) == ck) ++hi;
        uint32_t icol = 0, cf = my_first;
        bool up = false, down = false;
        for (uint32_t t = lo; t < hi; ++t) {
            if (t == i) continue;
            const uint32_t tf = first[t];
            cf = min(cf, tf);
            icol += (tf < my_first) ? 1u : 0u;
            const int tz = (int)(key[t] & 0x3FFFFFu) - (1 << 21);
            if (tz == za || tz == zb) {
                const uint32_t tc = count[t];
                const bool visited = tf < my_first && tc >= (uint32_t)P.min_points;
                const float oz = visited ? node_mean_z(tc, sums[9 * (size_t)t + 2], axis_centre(tz, P.oz, P.z_len)) : 0.f;
                const bool far = fabsf(oz - cz) > P.slope_interval;
                if (tz == za) up = up || far; else down = down || far;
            }
        }
        if (fl & 1u) {
            bool slope = true;
            if (P.demand == 0) slope = !up; else down = false;
            if (slope) { fl |= 2u; if (down) fl |= 4u; ++my_slopes; }
        }
        row.count = my_count; row.first = my_first; row.flags = fl;
        for (int j = 0; j < 3; ++j) row.mean[j] = 0.f;
        for (int j = 0; j < 6; ++j) row.scatter[j] = 0.0;
        if (fl & 1u) {
            double sm[9];
            for (int j = 0; j < 9; ++j) sm[j] = sums[9 * (size_t)i + j];
            const double c[3] = {axis_centre(row.sx, P.ox, P.grid_len), axis_centre(row.sy, P.oy, P.grid_len), cz_centre};
            node_moments(my_count, sm, c, row.mean, row.scatter);
        }
        row.col_first = cf; row.idx_in_col = icol; row.ncol = hi - lo;
        stage[i] = row;
        ord_cf[i] = cf;
        ord_idx[i] = icol;
        if (icol == 0) {
            if ((uint64_t)(cf >> 5) < words) note_column(O, cf, row.ncol);
            else atomicAdd(&pc->index_overflow, 1u);
            ++my_cols;
        }
    }
    if (my_slopes) atomicAdd(&s_slopes, my_slopes);
    if (my_cols) atomicAdd(&s_cols, my_cols);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_slopes) atomicAdd(&cnt->num_slopes, s_slopes);
        if (s_cols) atomicAdd(&cnt->num_columns, s_cols);
    }
}

}  // namespace gndt
