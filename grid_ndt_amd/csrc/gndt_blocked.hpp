// gndt_blocked.hpp — k_bucket_blocked: the bucket kernel for clouds whose occupied key range is a dense, evenly filled box.
//
// The hashed bucket kernel (gndt_bucket3.hpp) spends 25 of its 115 us of accumulate time on the bench scene FINDING a record's node
// (index window, fingerprint, key confirm) and 38 us grouping the nodes into columns afterwards (hash table of columns, per-column
// arrays: ~25 dependent LDS round trips per bucket) — profiles/r06_ablation.txt 1.  Both are the price of buckets that hold an
// arbitrary set of columns.  If a bucket is instead a spatial BLOCK of 2^shx x 2^shy columns x 2^shz levels = 512 nodes
// (GridParams::blk), a node's slot is a function of its key,
//        slot = (cz - z0) << (shx + shy) | (cy - y0 & mask_y) << shx | (cx - x0 & mask_x),
// the accumulate loop is key -> slot -> eleven atomics, and a node's column is the 2^shz slots that differ in the level bits: the
// column phases are eight independent LDS reads.  What it needs: every column of a block in ONE bucket and blocks of similar fill —
// i.e. a box of bounded height, evenly filled (the bench scene; a levelled site; not a LiDAR sweep, whose hottest hashed bucket is
// already 20 x the mean).  The host takes this kernel when the map of the previous build on the handle says so (partition_launch); a
// record that does not belong to its bucket's block (the cloud outgrew the box) raises PartCounters::blk_miss and the build is
// re-run with hashed buckets — the kernel is an optimisation of the same map, never another answer.
// Same outputs as k_bucket_direct: RawNode staging rows (a column's rows adjacent, first-seen order), ord_cf / ord_idx, the columns'
// votes in the bitmap / word weights, node / column / slope counts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_bucket3.hpp"

namespace gndt {

struct BlockedLds {            // 47 KB: three workgroups per CU
#if GNDT_BLOCKED_FIXED
    unsigned long long sum[9][512];    // FIXED-POINT sums (two's complement): see the accumulate loop
#else
    double sum[9][512];
#endif
    uint32_t cnt[512];
    uint32_t first[512];
    uint2 fz[512];             // {first-seen index (0xFFFFFFFF: no node in this slot), fp32 mean z of a node that has statistics, else 0}: what a
                               //   node's look at its column reads of the others, one 8-byte load per level
    uint32_t cpre[512];        // per COLUMN (the first 2^(shx+shy) entries): first row of the column inside the bucket
    uint32_t wave_tot[8];
    uint32_t n_nodes, n_cols, n_slopes, stage_base, err_range, miss, miss2;      // (miss2: raised after the accumulate phase — a word of its own, the first is being read then)
};

// Fixed-point contributions (-DGNDT_BLOCKED_FIXED=1; built, parity-green, measured, NOT the default).  In this kernel the LDS array, not
// the vector port, is the busy one (63 % against 34 %: profiles/r06_ablation.txt 8) and `ds_add_u64` takes half the time of `ds_add_f64` in
// isolation (10.2 against 19.9 clocks per wave-instruction at random slots, profiles/r03_lds_atomic_rates.txt) — so the nine sums were
// kept as 64-bit integers: a contribution x becomes round(x * 2^k) by ONE fused multiply-add against 1.5 * 2^52 (its low mantissa bits
// ARE the integer for |x * 2^k| < 2^51), the sums are order-independent (bit-identical statistics run to run).  Scales: offsets from the
// cell centre are below half a cell < 2^e, first moments take 2^(38 - e), second moments 2^(38 - 2 e); a node may hold 2^23 points
// before a sum could pass 2^61 (beyond: the bucket is a miss).  Measured in one call, twice: bucket kernel 124.0 | 124.0 us against
// 121.6 | 120.2 with fp64 sums, the accumulate stamps equal (33.6-34.2 k cycles) — the atomic unit's instruction rate is not what the
// 63 % are made of (bank conflicts and the queue behind them are) — and k_emit_rows 45-46 us against 34.5 (the quantised sums send more
// nodes through the eigen-solver's slow start).  Fp64 sums stay.
#ifndef GNDT_BLOCKED_FIXED
#define GNDT_BLOCKED_FIXED 0      // 1: fixed-point sums (A/B)
#endif
constexpr double kFixMagic = 6755399441055744.0;      // 1.5 * 2^52
constexpr uint32_t kFixMaxCount = 1u << 23;
__device__ __forceinline__ unsigned long long fix_round(double x, double scale) {
    const double t = fma(x, scale, kFixMagic);
    return (unsigned long long)__double_as_longlong(t) - (unsigned long long)__double_as_longlong(kFixMagic);
}

// One workgroup per bucket (the hardware's dynamic scheduling), three resident per CU.  The staging rows of a bucket are reserved with one
// memory-side atomic whose answer takes ~3 us; to have it in time the bucket's node count is known the moment the accumulate phase
// ends — a thread counts the slots it was the FIRST to add to (the count's atomic returns the old value, looked at one iteration
// later) — and the answer travels while the columns are worked out.  (Built and measured on the way, profiles/r06_ablation.txt 8: the count
// after the accumulate phase: ~5 k of 58 k cycles per bucket waiting; the rows of a bucket written behind the NEXT bucket's
// accumulate phase from registers: slower — the held row spills, and every vector-memory wait that follows stores waits for them.)
template <int T>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(6, 6))) k_bucket_blocked(const float4* __restrict__ recs, BucketRanges ranges, uint32_t num_buckets, GridParams P,
                                                      RawNode* __restrict__ stage, uint32_t stage_cap,
                                                      uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx, ColumnOrder O,
                                                      Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                      unsigned long long* __restrict__ dbg) {
    static_assert(T == 512, "one table slot per thread");
    __shared__ BlockedLds L;
    const int tid = threadIdx.x;
    const BlockMap K = P.blk;
    const int sh_xy = K.shx + K.shy;
    const uint32_t col_mask = (1u << sh_xy) - 1u, n_levels = 1u << K.shz;
    const double hx = 0.5 * (double)P.grid_len, hz = 0.5 * (double)P.z_len;
    const double ox = (double)P.ox, oy = (double)P.oy, oz = (double)P.oz;
    if (blockIdx.x == 0 && tid == 0) { const uint32_t e = pc->l1_err; if (e) atomicAdd(&cnt->err_key_range, e); }   // (FoldClear, gndt_partition.hpp)
    const int fix_e = ilogb(fmax(hx, hz)) + 1;                  // half a cell < 2^fix_e on every axis
    const double scale1 = ldexp(1.0, 38 - fix_e), scale2 = ldexp(1.0, 38 - 2 * fix_e);
    const double inv1 = ldexp(1.0, fix_e - 38), inv2 = ldexp(1.0, 2 * fix_e - 38);
    for (uint32_t bucket = blockIdx.x; bucket < num_buckets; bucket += gridDim.x) {
#define GNDT_STAMPB(k) do { if (dbg && tid == 0) dbg[(size_t)bucket * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
        uint32_t lo, hi;
        bucket_range(ranges, bucket, lo, hi);
        GNDT_STAMPB(0);
        // this bucket's block: contiguous indices bx0 .. bx0 + 2^shx - 1 (x), by0 .. (y), z0 .. z0 + 2^shz - 1 (levels)
        const int bx0 = K.x0 + (int)((bucket / (uint32_t)K.ny) << K.shx), by0 = K.y0 + (int)((bucket % (uint32_t)K.ny) << K.shy);
        {
            const int s = tid;
#pragma unroll
            for (int j = 0; j < 9; ++j) L.sum[j][s] = 0;
            L.cnt[s] = 0u; L.first[s] = 0xFFFFFFFFu;
            if (tid == 0) { L.n_nodes = 0; L.n_cols = 0; L.n_slopes = 0; L.stage_base = 0; L.err_range = 0; L.miss = 0; L.miss2 = 0; }
        }
        lds_barrier();
        GNDT_STAMPB(1);
        // ---- accumulate: one record per thread and iteration, the next one's load in flight ----
        const uint32_t n_rec = hi - lo, iters = (n_rec + (uint32_t)T - 1u) / (uint32_t)T;
        float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lo < hi) nxt = recs[min(lo + (uint32_t)tid, hi - 1u)];
        uint32_t my_new = 0, old_cnt = 1u;          // slots this thread was the first to add to (old_cnt: the last atomic's answer, looked at an iteration later)
        for (uint32_t it = 0; it < iters; ++it) {
            const uint32_t mine = lo + it * (uint32_t)T + (uint32_t)tid;
            const float4 rec = nxt;
            bool use = mine < hi;
            if (it + 1u < iters) nxt = recs[min(mine + (uint32_t)T, hi - 1u)];
            bool und = false;
            float fx = axis_ceil_try(rec.x, P.ox, P.inv_grid, und);
            float fy = axis_ceil_try(rec.y, P.oy, P.inv_grid, und);
            float fz = axis_ceil_try(rec.z, P.oz, P.inv_z, und);
            if (und) {                                 // (rare: within ~2 ulp of a cell border the reference's own divide decides)
                fx = ceilf(fabsf(rec.x - P.ox) / P.grid_len);
                fy = ceilf(fabsf(rec.y - P.oy) / P.grid_len);
                fz = ceilf(fabsf(rec.z - P.oz) / P.z_len);
            }
            bool ok = true;
            int sx, sy, sz;
            double v0, v1, v2;
            axis_index_offset(rec.x, P.ox, fx, (float)kMaxXY, hx, ox, ok, sx, v0);
            axis_index_offset(rec.y, P.oy, fy, (float)kMaxXY, hx, oy, ok, sy, v1);
            axis_index_offset(rec.z, P.oz, fz, (float)kMaxZ, hz, oz, ok, sz, v2);
            if (use && !ok) { atomicAdd(&L.err_range, 1u); use = false; }      // |nz| beyond the key range (x, y: the partition)
            const uint32_t lx = (uint32_t)(contiguous_index(sx) - bx0), ly = (uint32_t)(contiguous_index(sy) - by0),
                           lz = (uint32_t)(contiguous_index(sz) - K.z0);
            if (use && ((lx >> K.shx) | (ly >> K.shy) | (lz >> K.shz)) != 0u) { L.miss = 1u; use = false; }      // not this block's: the build is re-run hashed
            const uint32_t s = (lz << sh_xy) | (ly << K.shx) | lx;
            const uint32_t iw = __float_as_uint(rec.w);
            uint32_t cn = 1u, cf = iw;
            double w0 = v0, w1 = v1, w2 = v2;
            if (__any((iw & kWeight64Flag) != 0u)) {              // (wave-uniform) weighted records: 64 or 512 identical points in one
                cn = record_weight(iw); cf = record_index(iw);
                const double wf = (double)cn;
                w0 = wf * v0; w1 = wf * v1; w2 = wf * v2;
            }
            my_new += old_cnt == 0u ? 1u : 0u;          // (the answer of the iteration before)
            old_cnt = 1u;
            if (use) {
                old_cnt = atomicAdd(&L.cnt[s], cn);
#if GNDT_BLOCKED_FIXED
                atomicAdd(&L.sum[0][s], fix_round(w0, scale1)); atomicAdd(&L.sum[1][s], fix_round(w1, scale1)); atomicAdd(&L.sum[2][s], fix_round(w2, scale1));
                atomicAdd(&L.sum[3][s], fix_round(w0 * v0, scale2)); atomicAdd(&L.sum[4][s], fix_round(w0 * v1, scale2)); atomicAdd(&L.sum[5][s], fix_round(w0 * v2, scale2));
                atomicAdd(&L.sum[6][s], fix_round(w1 * v1, scale2)); atomicAdd(&L.sum[7][s], fix_round(w1 * v2, scale2)); atomicAdd(&L.sum[8][s], fix_round(w2 * v2, scale2));
#else
                atomicAdd(&L.sum[0][s], w0); atomicAdd(&L.sum[1][s], w1); atomicAdd(&L.sum[2][s], w2);
                atomicAdd(&L.sum[3][s], w0 * v0); atomicAdd(&L.sum[4][s], w0 * v1); atomicAdd(&L.sum[5][s], w0 * v2);
                atomicAdd(&L.sum[6][s], w1 * v1); atomicAdd(&L.sum[7][s], w1 * v2); atomicAdd(&L.sum[8][s], w2 * v2);
#endif
                atomicMin(&L.first[s], cf);
            }
        }
        my_new += old_cnt == 0u ? 1u : 0u;
        {   // the bucket's node count: per wave, one LDS atomic each
            uint32_t w = my_new;
            for (int off = 32; off > 0; off >>= 1) w += (uint32_t)__shfl_down((int)w, off, 64);
            if ((tid & 63) == 0 && w) atomicAdd(&L.n_nodes, w);
        }
        lds_barrier();
        GNDT_STAMPB(2);
        if (L.miss) {                                  // (uniform)
            if (tid == 0) atomicAdd(&pc->blk_miss, 1u);
            lds_barrier();
            continue;
        }
        const uint32_t M = L.n_nodes;
        // the staging rows: asked for now, the answer is awaited in front of the rows (thread T - 1 keeps it in a register until then)
        uint32_t stage_base_reg = 0;
        if (tid == T - 1 && M) stage_base_reg = atomicAdd(&cnt->num_nodes, M);
        if (tid == 0 && L.err_range) atomicAdd(&cnt->err_key_range, L.err_range);
        // ---- the bucket's nodes: slot s in thread s; its statistics go to registers, {first-seen, mean z} to LDS ----
        const uint32_t s = (uint32_t)tid;
        const uint32_t my_n = L.cnt[s], my_first = L.first[s];
        double sums[9];
#pragma unroll
#if GNDT_BLOCKED_FIXED
        for (int j = 0; j < 9; ++j) sums[j] = (double)(long long)L.sum[j][s] * (j < 3 ? inv1 : inv2);
#else
        for (int j = 0; j < 9; ++j) sums[j] = L.sum[j][s];
#endif
        const bool live = my_n != 0u;
#if GNDT_BLOCKED_FIXED
        if (my_n > kFixMaxCount) L.miss2 = 1u;                  // (a node of more points than the fixed-point sums are sized for: hashed buckets)
#endif
        const uint32_t col = s & col_mask, lz = s >> sh_xy;
        // signed indices of this slot's node
        const int cxi = bx0 + (int)(col & ((1u << K.shx) - 1u)), cyi = by0 + (int)(col >> K.shx), czi = K.z0 + (int)lz;
        const int nsx = cxi >= 0 ? cxi + 1 : cxi, nsy = cyi >= 0 ? cyi + 1 : cyi, nsz = czi >= 0 ? czi + 1 : czi;
        const float cz = (live && my_n >= (uint32_t)P.min_points) ? node_mean_z(my_n, sums[2], axis_centre(nsz, P.oz, P.z_len)) : 0.f;
        L.fz[s] = make_uint2(my_first, __float_as_uint(cz));
        lds_barrier();
        if (L.miss2) {                                 // (uniform; nothing of this bucket has left the workgroup but its reservation: the build is re-run)
            if (tid == 0) atomicAdd(&pc->blk_miss, 1u);
            lds_barrier();
            continue;
        }
        // ---- the column of every node: the slots that differ in the level bits — independent 8-byte reads, no search.  The thread of a
        //      column's level-0 slot walks the column whether that slot holds a node or not: its node count is what the prefix over the
        //      columns (first row of every column inside the bucket) is made of ----
        uint32_t icol = 0, ncol = 0, cfirst = 0xFFFFFFFFu;
        bool up = false, down = false;
        if (live || s <= col_mask) {
            for (uint32_t k0 = 0; k0 < n_levels; k0 += 4) {
                uint2 tt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tt[j] = L.fz[min(((k0 + (uint32_t)j) << sh_xy) | col, 511u)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t k = k0 + (uint32_t)j, tf = tt[j].x;
                    if (k >= n_levels || tf == 0xFFFFFFFFu) continue;
                    ++ncol;
                    cfirst = min(cfirst, tf);
                    if (k == lz) continue;
                    icol += (tf < my_first) ? 1u : 0u;
                    // OcNode::isSlope (map2D.h:66-108): the node one level up / down counts with its centroid only if it was seen
                    // earlier AND has statistics (its mean z is 0 below min_points), else with 0.0f.  In contiguous level indices "one
                    // level up" is k == lz + 1 — level_above / level_below skip the index 0 that does not exist.
                    if (k == lz + 1u || k + 1u == lz) {
                        const float oz2 = (tf < my_first) ? __uint_as_float(tt[j].y) : 0.f;
                        const bool far = fabsf(oz2 - cz) > P.slope_interval;
                        if (k == lz + 1u) up = far; else down = far;
                    }
                }
            }
        }
        {   // exclusive prefix of the columns' node counts: by shuffles inside a wave; a column of the second wave (blocks of 128
            // columns) adds the first wave's total when it reads its entry
            const uint32_t my_col_nodes = s <= col_mask ? ncol : 0u;
            uint32_t incl = my_col_nodes;
            const int lane = tid & 63;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
            if (s <= col_mask) L.cpre[s] = incl - my_col_nodes;
            if (lane == 63) L.wave_tot[tid >> 6] = incl;
            if (s <= col_mask && my_col_nodes) atomicAdd(&L.n_cols, 1u);
            if (tid == T - 1) L.stage_base = stage_base_reg;      // (the reservation's answer: waited for here, by one thread)
            lds_barrier();
        }
        GNDT_STAMPB(3);
        const uint32_t sbase = L.stage_base;
        if (sbase + M > stage_cap) {                   // uniform: the staging rows ran out, the build is re-run with more
            if (tid == 0 && M) atomicAdd(&pc->stage_overflow, M);
            lds_barrier();
            continue;
        }
        uint32_t my_slopes = 0;
        if (live) {
            uint32_t fl = (my_n >= (uint32_t)P.min_points) ? 1u : 0u;
            if (fl) {
                bool slope = true;
                if (P.demand == 0) slope = !up; else down = false;
                if (slope) { fl |= 2u; if (down) fl |= 4u; ++my_slopes; }
            }
            RawNode row;
            row.key = pack_key(nsx, nsy, nsz); row.count = my_n; row.first = my_first;
#pragma unroll
            for (int j = 0; j < 9; ++j) row.sum[j] = sums[j];
            row.info = fl | (icol << 3);
            row.ncol = ncol;
            uint32_t cbase = L.cpre[col];
            for (uint32_t w = 0; w < (col >> 6); ++w) cbase += L.wave_tot[w];      // (columns 64 .. 127 of a 128-column block: behind the first wave's)
            const uint32_t dst = sbase + cbase + icol;
            stage[dst] = row;
            ord_cf[dst] = cfirst;
            ord_idx[dst] = icol ? icol : (kOrdHeadFlag | ncol);       // (a column's first row carries the column's size)
            if (icol == 0) note_column(O, cfirst, ncol);
        }
        if (my_slopes) atomicAdd(&L.n_slopes, my_slopes);
        lds_barrier();
        if (tid == 0) {
            // (ONE atomic, on the partition counters' line — PartCounters::cols_slopes: not on the line the row reservations wait on)
            if (L.n_cols | L.n_slopes) atomicAdd(&pc->cols_slopes, ((unsigned long long)L.n_slopes << 32) | (unsigned long long)L.n_cols);
        }
        GNDT_STAMPB(4);
        lds_barrier();          // (the table is re-initialised by the next bucket)
#undef GNDT_STAMPB
    }
}

// The box a finished map occupies, in contiguous indices: min / max of cx, cy, cz over its rows and the largest node (what decides whether
// the next build of a cloud like it may take blocked buckets).  64 workgroups, one LDS reduction each, twelve memory-side atomics per
// workgroup; out[0..5] = min x, y, z (as biased uint32: + 2^30), max x, y, z, out[6] = largest count — initialised by the caller.
static __global__ void __launch_bounds__(1024) k_key_extent(const int32_t* __restrict__ sx, const int32_t* __restrict__ sy, const int32_t* __restrict__ sz,
                                                            const uint32_t* __restrict__ count, uint32_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t red[7];
    if (threadIdx.x < 3) red[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 7) red[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u}, big = 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t c[3] = {(uint32_t)(contiguous_index(sx[i]) + (1 << 30)), (uint32_t)(contiguous_index(sy[i]) + (1 << 30)),
                               (uint32_t)(contiguous_index(sz[i]) + (1 << 30))};
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], c[k]); mx[k] = max(mx[k], c[k]); }
        big = max(big, count[i]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = min(mn[k], (uint32_t)__shfl_down((int)mn[k], off, 64));
            mx[k] = max(mx[k], (uint32_t)__shfl_down((int)mx[k], off, 64));
        }
    }
    for (int off = 32; off > 0; off >>= 1) big = max(big, (uint32_t)__shfl_down((int)big, off, 64));
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { atomicMin(&red[k], mn[k]); atomicMax(&red[3 + k], mx[k]); }
        atomicMax(&red[6], big);
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&out[threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 7) atomicMax(&out[threadIdx.x], red[threadIdx.x]);
}

}  // namespace gndt
