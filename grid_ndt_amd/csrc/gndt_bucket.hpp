// gndt_bucket.hpp — k_bucket_build2: one workgroup per bucket, second generation.
//
// The first bucket kernel (gndt_partition.hpp: k_bucket_build) spent ~650 lane-instructions per point:
// nine fp64 LDS atomics + products + centres per point, an O(M^2) ranking loop and hash probes for the
// slope test.  It is VALU-issue bound (in-kernel stamps).  This version does the same job
// with ~1/3 of the instructions:
//
//   per chunk of up to CH points of the bucket
//     A  classify : key -> LDS table slot (probe / CAS), arrival rank in the slot by ONE returning u32 atomic
//     B  scatter  : exclusive scan of the slot counts, points written to a slot-sorted SoA image in LDS
//     C  reduce   : each thread walks K consecutive sorted points; runs of one node accumulate in REGISTERS
//                   (fp64 Sum v, Sum v v^T about the node centre) and are flushed with one set of fp64
//                   LDS atomics per run (~1.5 runs per thread instead of 9 atomics per point)
//   then, on the finished table
//     D  columns  : column table + a linked list of each column's nodes
//     E  labels   : slope test and index-in-column by walking the (short) column list: no hash probes;
//                   the node's staging row is written in the same pass
//     F  order    : column base = sum of the sizes of the columns first seen earlier (loop over the bucket's
//                   columns, not its nodes); row = base + index in column
//     F  emit     : mean + fp64 scatter -> 128-B staging row; bitmap bit per column-first index
//
// Semantics are those of k_bucket_build (same gndt_math.hpp arithmetic); tests run both.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_partition.hpp"

namespace gndt {

template <int H, int CH>
struct BucketLds2 {
    // persistent across chunks
    unsigned long long key[H];
    double sum[9][H];
    uint32_t cnt[H];
    uint32_t first[H];
    uint32_t ccur[H];     // per-chunk arrival counter, then exclusive start in the sorted image
    union {
        struct {          // chunk phases A-C: slot-sorted SoA image of the chunk
            float x[CH], y[CH], z[CH];
            uint32_t idx[CH];
            uint16_t slot[CH];
        } pts;
        struct {          // phases D-G
            unsigned long long ckey[H];
            uint32_t cfirst[H], ccnt[H], chead[H];
            uint32_t next[H];       // next node (slot) of the same column, 0xFFFFFFFF ends
            float mean_z[H];
            uint32_t flags[H];      // bits 0..2 GNDT_FLAG_*, bits 8.. column slot
        } fin;
    } u;
    uint32_t wave_tot[16];
    uint32_t n_nodes, n_cols, n_clist, n_rows, n_slopes, stage_base, overflow;
};

// Compact per-node statistics (the gndt_stats layout): what a shard of a multi-GPU build hands to the exchange.
struct StatsOut {
    uint64_t* key; double* sums; uint32_t* count; uint32_t* first;
};

// STATS = false: the bucket's nodes leave as staging rows (labels, moments) for the ordering + emit kernels.
// STATS = true : they leave as additive statistics (key, 9 sums, count, first index) and nothing else is done:
//                the shard's contribution to a global map (gndt_shard_stats_device).
template <int T, int H, int CH, bool STATS>
__device__ __forceinline__ void bucket_build_one(BucketLds2<H, CH>& L, const uint32_t bucket, const uint32_t num_buckets,
                                                 const float4* __restrict__ recs, const uint32_t* __restrict__ range_lo,
                                                 const uint32_t* __restrict__ range_hi, const GridParams& P,
                                                 StageRow* __restrict__ stage, uint32_t stage_cap,
                                                 uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                 const ColumnOrder& O,
                                                 Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                 unsigned long long* __restrict__ dbg, const StatsOut& so,
                                                 float4 (&pre)[CH / T], uint32_t& pre_bucket, uint32_t& pre_cbeg) {
    static_assert(CH % T == 0, "chunk must be a multiple of the block");
    static_assert(H % T == 0 || T % H == 0, "slots vs threads");
    constexpr int PER = CH / T;                 // points per thread and chunk
    constexpr int SPT = (H + T - 1) / T;        // slots per thread in the per-slot loops
    constexpr int kFill = (H * 25) / 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#define GNDT_STAMP(k) do { if (dbg && tid == 0) dbg[(size_t)bucket * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    GNDT_STAMP(0);
    for (int s = tid; s < H; s += T) {
        L.key[s] = kEmptyKey;
#pragma unroll
        for (int j = 0; j < 9; ++j) L.sum[j][s] = 0.0;
        L.cnt[s] = 0; L.first[s] = 0xFFFFFFFFu; L.ccur[s] = 0;
    }
    if (tid == 0) { L.n_nodes = 0; L.n_cols = 0; L.n_clist = 0; L.n_rows = 0; L.n_slopes = 0; L.stage_base = 0; L.overflow = 0; }
    __syncthreads();
    GNDT_STAMP(1);

    const uint32_t lo = range_lo[bucket], hi = range_hi[bucket];   // exact path: bucket_base[b], bucket_base[b+1]
    const uint32_t nb = bucket + gridDim.x;                        // this workgroup's next bucket
    uint32_t nlo = 0, nhi = 0;
    if (nb < num_buckets) { nlo = range_lo[nb]; nhi = range_hi[nb]; }
    unsigned long long acc_t[4] = {0, 0, 0, 0}, t_prev = 0;     // diagnostic: load / classify / scatter / reduce, all chunks
#define GNDT_LAP(k) do { if (dbg && tid == 0) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); acc_t[k] += t_now - t_prev; t_prev = t_now; } } while (0)
    if (dbg && tid == 0) t_prev = __builtin_amdgcn_s_memtime();
    for (uint32_t cbeg = lo; cbeg < hi; cbeg += CH) {
        const uint32_t nchunk = min((uint32_t)CH, hi - cbeg);
        // ---- A: classify ----
        float4 rec[PER];
        uint32_t tag[PER];                      // slot << 16 | arrival rank (CH <= 65536)
        if (pre_bucket == bucket && pre_cbeg == cbeg) {           // (uniform) the records were loaded a chunk ago
#pragma unroll
            for (int j = 0; j < PER; ++j) rec[j] = pre[j];
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t off = (uint32_t)j * T + tid;
                rec[j] = (off < nchunk) ? recs[cbeg + off] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        {   // issue the loads of the chunk this workgroup processes next: the rest of this bucket, or its next bucket
            uint32_t pb = bucket, pc = cbeg + CH, pe = hi;
            if (pc >= hi) { pb = nb; pc = nlo; pe = nhi; }
            pre_bucket = 0xFFFFFFFFu;
            if (pb < num_buckets && pc < pe) {
                const uint32_t pn = min((uint32_t)CH, pe - pc);
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    const uint32_t off = (uint32_t)j * T + tid;
                    pre[j] = (off < pn) ? recs[pc + off] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                pre_bucket = pb; pre_cbeg = pc;
            }
        }
        if (dbg && tid == 0) { float keep = 0.f; for (int j = 0; j < PER; ++j) keep += rec[j].x; if (keep == 1.2345e-30f) dbg[1] = 0; }
        GNDT_LAP(0);
        // Staged so that the PER independent points overlap their latencies: all keys and hashes (VALU), then
        // all first probes (plain LDS loads), then the rare slow paths, then all arrival-rank atomics.
        uint64_t pkey[PER];
        uint32_t ph[PER];
        bool kok[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const PointKey k = point_key_fast(rec[j].x, rec[j].y, rec[j].z, P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
            kok[j] = k.ok;                       // (x, y were range-checked by the partition; z is checked here)
            pkey[j] = pack_key(k.sx, k.sy, k.sz);
            ph[j] = node_slot_hash(column_hash(k.sx, k.sy), k.sz) & (uint32_t)(H - 1);
        }
        unsigned long long k0[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) k0[j] = L.key[ph[j]];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t off = (uint32_t)j * T + tid;
            tag[j] = 0xFFFFFFFFu;
            if (off < nchunk && !kok[j]) atomicAdd(&cnt->err_key_range, 1u);
            if (off < nchunk && kok[j]) {
                uint32_t s = ph[j];
                if (k0[j] != pkey[j]) s = lds_find_or_insert<H>(L.key, ph[j], pkey[j], &L.n_nodes);
                if (s >= (uint32_t)H) L.overflow = 1;
                else tag[j] = s << 16;
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (tag[j] != 0xFFFFFFFFu) tag[j] |= atomicAdd(&L.ccur[tag[j] >> 16], 1u);
        __syncthreads();
        GNDT_LAP(1);
        if (L.overflow || L.n_nodes > (uint32_t)kFill) {     // uniform
            if (tid == 0) atomicAdd(&pc->lds_overflow, 1u);
            return;
        }
        // ---- B: exclusive scan of the chunk counts over the slots, then scatter into the sorted image ----
        {
            uint32_t c[SPT], tot = 0;
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int s = tid * SPT + q; c[q] = (s < H) ? L.ccur[s] : 0u; tot += c[q]; }
            uint32_t incl = tot;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) L.wave_tot[wave] = incl;
            __syncthreads();
            uint32_t run = incl - tot;
            for (int w = 0; w < wave; ++w) run += L.wave_tot[w];
#pragma unroll
            for (int q = 0; q < SPT; ++q) { const int s = tid * SPT + q; if (s < H) { L.ccur[s] = run; run += c[q]; } }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (tag[j] != 0xFFFFFFFFu) {
                const uint32_t s = tag[j] >> 16;
                const uint32_t pos = L.ccur[s] + (tag[j] & 0xFFFFu);
                L.u.pts.x[pos] = rec[j].x; L.u.pts.y[pos] = rec[j].y; L.u.pts.z[pos] = rec[j].z;
                L.u.pts.idx[pos] = __float_as_uint(rec[j].w);
                L.u.pts.slot[pos] = (uint16_t)s;
            }
        }
        __syncthreads();
        GNDT_LAP(2);
        // ---- C: run-length accumulation over the sorted image ----
        {
            const uint32_t p0 = (uint32_t)tid * PER;
            // preload this thread's K consecutive sorted points (stride-K dword reads are conflict-free for odd K)
            float qx[PER], qy[PER], qz[PER];
            uint32_t qi[PER], qs[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t p = p0 + j;
                const bool ok = p < nchunk;
                qs[j] = ok ? (uint32_t)L.u.pts.slot[p] : 0xFFFFFFFFu;
                qx[j] = ok ? L.u.pts.x[p] : 0.f; qy[j] = ok ? L.u.pts.y[p] : 0.f; qz[j] = ok ? L.u.pts.z[p] : 0.f;
                qi[j] = ok ? L.u.pts.idx[p] : 0xFFFFFFFFu;      // bit 31: the record stands for 64 identical points
            }
            uint32_t cur = 0xFFFFFFFFu, rn = 0, rfirst = 0xFFFFFFFFu;
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
            double c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t s = qs[j];
                if (s != cur) {
                    if (cur != 0xFFFFFFFFu) {
                        atomicAdd(&L.sum[0][cur], a0); atomicAdd(&L.sum[1][cur], a1); atomicAdd(&L.sum[2][cur], a2);
                        atomicAdd(&L.sum[3][cur], a3); atomicAdd(&L.sum[4][cur], a4); atomicAdd(&L.sum[5][cur], a5);
                        atomicAdd(&L.sum[6][cur], a6); atomicAdd(&L.sum[7][cur], a7); atomicAdd(&L.sum[8][cur], a8);
                        atomicAdd(&L.cnt[cur], rn); atomicMin(&L.first[cur], rfirst);
                    }
                    cur = s; rn = 0; rfirst = 0xFFFFFFFFu;
                    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = a8 = 0.0;
                    if (s != 0xFFFFFFFFu) {
                        int sx, sy, sz;
                        unpack_key(L.key[s], sx, sy, sz);
                        c0 = axis_centre(sx, P.ox, P.grid_len); c1 = axis_centre(sy, P.oy, P.grid_len); c2 = axis_centre(sz, P.oz, P.z_len);
                    }
                }
                if (s != 0xFFFFFFFFu) {
                    const uint32_t wn = record_weight(qi[j]);
                    const double w = (double)wn;
                    const double v0 = (double)qx[j] - c0, v1 = (double)qy[j] - c1, v2 = (double)qz[j] - c2;
                    const double w0 = w * v0, w1 = w * v1, w2 = w * v2;     // exact: w is a power of two
                    a0 += w0; a1 += w1; a2 += w2;
                    // fused multiply-add on purpose: one rounding per term (the sums are order-free anyway)
                    a3 = fma(w0, v0, a3); a4 = fma(w0, v1, a4); a5 = fma(w0, v2, a5);
                    a6 = fma(w1, v1, a6); a7 = fma(w1, v2, a7); a8 = fma(w2, v2, a8);
                    rn += wn;
                    rfirst = min(rfirst, record_index(qi[j]));
                }
            }
            if (cur != 0xFFFFFFFFu) {
                atomicAdd(&L.sum[0][cur], a0); atomicAdd(&L.sum[1][cur], a1); atomicAdd(&L.sum[2][cur], a2);
                atomicAdd(&L.sum[3][cur], a3); atomicAdd(&L.sum[4][cur], a4); atomicAdd(&L.sum[5][cur], a5);
                atomicAdd(&L.sum[6][cur], a6); atomicAdd(&L.sum[7][cur], a7); atomicAdd(&L.sum[8][cur], a8);
                atomicAdd(&L.cnt[cur], rn); atomicMin(&L.first[cur], rfirst);
            }
        }
        __syncthreads();
        for (int s = tid; s < H; s += T) L.ccur[s] = 0;     // next chunk (nobody reads ccur before the next barrier)
        __syncthreads();
        GNDT_LAP(3);
    }
#undef GNDT_LAP
    if (dbg && tid == 0) for (int k = 0; k < 4; ++k) dbg[(size_t)bucket * 16 + 8 + k] = acc_t[k];
    GNDT_STAMP(2);

    // Reserve the staging rows now: the memory-side atomic's round trip hides behind phases D-F.
    uint32_t stage_base_reg = 0;
    const uint32_t M = L.n_nodes;
    if (tid == T - 1) stage_base_reg = atomicAdd(&cnt->num_nodes, M);

    if constexpr (STATS) {
        if (tid == T - 1) L.stage_base = stage_base_reg;
        __syncthreads();
        const uint32_t sbase = L.stage_base;
        if (sbase + M > stage_cap) {               // uniform
            if (tid == 0) atomicAdd(&pc->stage_overflow, M);
            return;
        }
        for (int s = tid; s < H; s += T) {
            const uint64_t key = L.key[s];
            if (key == kEmptyKey) continue;
            const uint32_t dst = sbase + atomicAdd(&L.n_rows, 1u);
            so.key[dst] = key;
#pragma unroll
            for (int j = 0; j < 9; ++j) so.sums[9 * (size_t)dst + j] = L.sum[j][s];
            so.count[dst] = L.cnt[s];
            so.first[dst] = L.first[s];
        }
        return;
    }

    // ---- D: columns (the pts image is dead: its LDS now holds the column tables) ----
    for (int s = tid; s < H; s += T) { L.u.fin.ckey[s] = kEmptyKey; L.u.fin.cfirst[s] = 0xFFFFFFFFu; L.u.fin.ccnt[s] = 0; L.u.fin.chead[s] = 0xFFFFFFFFu; }
    __syncthreads();
    for (int s = tid; s < H; s += T) {
        const uint64_t key = L.key[s];
        if (key == kEmptyKey) continue;
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        const uint32_t n = L.cnt[s];
        uint32_t fl = 0;
        float mz = 0.f;
        if (n >= (uint32_t)P.min_points) { mz = node_mean_z(n, L.sum[2][s], axis_centre(sz, P.oz, P.z_len)); fl = 1u; }
        L.u.fin.mean_z[s] = mz;
        const uint32_t cs = lds_find_or_insert<H>(L.u.fin.ckey, column_hash(sx, sy) * 0x85EBCA77u >> 12, column_key(key), &L.n_cols);
        atomicMin(&L.u.fin.cfirst[cs], L.first[s]);
        atomicAdd(&L.u.fin.ccnt[cs], 1u);
        L.u.fin.next[s] = atomicExch(&L.u.fin.chead[cs], (uint32_t)s);
        L.u.fin.flags[s] = fl | (cs << 8);
    }
    if (tid == T - 1) L.stage_base = stage_base_reg;
    __syncthreads();
    GNDT_STAMP(3);
    const uint32_t base0 = L.stage_base;
    if (base0 + M > stage_cap) {               // uniform
        if (tid == 0) atomicAdd(&pc->stage_overflow, M);
        return;
    }
    if (tid == 0) atomicAdd(&cnt->num_columns, L.n_cols);

    // ---- E: slope labels (OcNode::isSlope, map2D.h:66-108) and index in column, by walking the column list ----
    uint32_t my_slopes = 0;
    for (int s = tid; s < H; s += T) {
        const uint64_t key = L.key[s];
        if (key == kEmptyKey) continue;
        uint32_t fl = L.u.fin.flags[s];
        const uint32_t cs = fl >> 8;
        const uint32_t my_first = L.first[s];
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        const int za = level_above(sz), zb = level_below(sz);
        const float cz = L.u.fin.mean_z[s];
        uint32_t icol = 0;
        bool up = false, down = false;
        for (uint32_t t = L.u.fin.chead[cs]; t != 0xFFFFFFFFu; t = L.u.fin.next[t]) {
            if (t == (uint32_t)s) continue;
            const uint32_t tf = L.first[t];
            icol += (tf < my_first) ? 1u : 0u;
            const int tz = (int)(L.key[t] & 0x3FFFFFu) - (1 << 21);
            if (tz == za || tz == zb) {
                const bool visited = tf < my_first && (L.u.fin.flags[t] & 1u);
                const float oz = visited ? L.u.fin.mean_z[t] : 0.f;
                const bool far = fabsf(oz - cz) > P.slope_interval;
                if (tz == za) up = up || far; else down = down || far;
            }
        }
        if (fl & 1u) {
            bool slope = true;
            if (P.demand == 0) slope = !up; else down = false;
            if (slope) { fl |= 2u; if (down) fl |= 4u; ++my_slopes; }
        }
        // ---- F: staging row, written straight away (rows of a bucket need no order among themselves: the
        //         global ordering kernels only use column first-seen and index in column) ----
        const uint32_t idx_in_col = icol;
        const uint32_t cf = L.u.fin.cfirst[cs];
        fl &= 7u;
        StageRow row;
        unpack_key(key, row.sx, row.sy, row.sz);
        row.count = L.cnt[s]; row.first = L.first[s]; row.flags = fl;
        for (int k = 0; k < 3; ++k) row.mean[k] = 0.f;
        for (int k = 0; k < 6; ++k) row.scatter[k] = 0.0;
        if (fl & 1u) {
            double sums[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) sums[j] = L.sum[j][s];
            const double c[3] = {axis_centre(row.sx, P.ox, P.grid_len), axis_centre(row.sy, P.oy, P.grid_len),
                                 axis_centre(row.sz, P.oz, P.z_len)};
            node_moments(row.count, sums, c, row.mean, row.scatter);
        }
        row.col_first = cf; row.idx_in_col = idx_in_col; row.ncol = L.u.fin.ccnt[cs];
        const uint32_t dst = base0 + atomicAdd(&L.n_rows, 1u);
        stage[dst] = row;
        ord_cf[dst] = cf;
        ord_idx[dst] = idx_in_col;
        if (idx_in_col == 0) note_column(O, cf, row.ncol);
    }
    // counters: aggregated in LDS, ONE memory-side atomic per bucket and counter (same-address global atomics
    // serialise at the memory side and slow every other request down with them)
    if (my_slopes) atomicAdd(&L.n_slopes, my_slopes);
    __syncthreads();
    if (tid == 0 && L.n_slopes) atomicAdd(&cnt->num_slopes, L.n_slopes);
    if (dbg) { GNDT_STAMP(4); GNDT_STAMP(5); }
    if (dbg) { __syncthreads(); GNDT_STAMP(6); }
#undef GNDT_STAMP
}


// Bucket b, b + gridDim.x, ... (one bucket per workgroup unless the grid is smaller than the bucket count); the records of
// the chunk a workgroup processes next are loaded into registers while it works on the current one.  128 VGPRs at most:
// two 512-thread workgroups per CU are what the LDS allows, and they need 4 waves per SIMD.
template <int T, int H, int CH, bool STATS = false>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bucket_build2(const float4* __restrict__ recs, const uint32_t* __restrict__ range_lo,
                                                     const uint32_t* __restrict__ range_hi, uint32_t num_buckets,
                                                     GridParams P, StageRow* __restrict__ stage, uint32_t stage_cap,
                                                     uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                     ColumnOrder O,
                                                     Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                     unsigned long long* __restrict__ dbg, StatsOut so) {
    __shared__ BucketLds2<H, CH> L;
    float4 pre[CH / T];
    uint32_t pre_bucket = 0xFFFFFFFFu, pre_cbeg = 0;
    for (uint32_t bucket = blockIdx.x; bucket < num_buckets; bucket += gridDim.x) {
        bucket_build_one<T, H, CH, STATS>(L, bucket, num_buckets, recs, range_lo, range_hi, P, stage, stage_cap, ord_cf, ord_idx,
                                          O, cnt, pc, dbg, so, pre, pre_bucket, pre_cbeg);
        __syncthreads();        // the LDS tables are re-initialised by the next bucket
    }
}

}  // namespace gndt
