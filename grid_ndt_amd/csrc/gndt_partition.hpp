// gndt_partition.hpp — strategy PARTITION: the whole build with every per-node atomic kept in LDS.
//
// The atomic path (gndt_kernels.hpp) spends its time in ~11 memory-side atomics per point (MI355X
// executes device-scope atomics at the memory side, ~20 G requests/s chip-wide).  Here the points are
// first partitioned by COLUMN hash into B buckets with a counting partition (two streaming passes), so
// that one workgroup owns every node of its bucket's columns and can keep their statistics, the slope
// labels and the in-bucket ordering in LDS:
//
//   k_part_hist      points -> bucket histogram per workgroup                      reads 12 B/pt
//   k_part_offsets   per-bucket exclusive scan over workgroups (+ bucket totals)
//   k_part_scatter   points -> {x,y,z,idx} records grouped by bucket               reads 12, writes 16 B/pt
//   k_bucket_build   one workgroup per bucket: LDS hash table of nodes, fp64 LDS atomics,
//                    column table, slope labels, bitonic sort by (column first-seen, node first-seen),
//                    mean/scatter/eigen -> 128-B staging rows                      reads 16 B/pt, writes 128 B/node
//   k_scan_*         bitmap of column-first point indices -> column rank; column sizes -> row offsets
//   k_order_*        destination row of every node (reference order), inverse permutation
//   k_emit_rows      staging rows -> SoA result in reference order                 reads 128, writes 76 B/node
//
// Reference semantics are the ones of gndt_kernels.hpp (same gndt_math.hpp arithmetic); only the data
// movement differs.  Anything that does not fit (LDS table overflow, staging overflow) raises a flag
// and the host re-runs the build on the atomic path, so results never depend on the strategy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_kernels.hpp"

namespace gndt {

constexpr int kPartThreads = 1024;   // k_part_hist / k_part_scatter
// k_bucket_build is a template on <threads, LDS node-table slots>; a bucket holding more than
// 0.78 * slots distinct nodes overflows (the host then re-runs on the atomic path).
constexpr int kScanChunk = 2048;     // elements per block in the two-level scans
constexpr int kScanThreads = 256;

struct PartCounters {
    uint32_t lds_overflow;     // buckets whose node table overflowed
    uint32_t stage_overflow;   // nodes that did not fit the staging rows
    uint32_t pad[2];
};

struct alignas(16) StageRow {   // 128 bytes = two 64-B lines, gathered whole by k_emit_rows
    int32_t sx, sy, sz;
    uint32_t count, first, flags;
    uint32_t col_first, idx_in_col, ncol;
    float mean[3];
    double scatter[6];          // fp64: the eigen-solve runs in k_emit_rows (chip-wide parallelism)
    uint32_t pad[8];
};
static_assert(sizeof(StageRow) == 128, "StageRow layout");

__host__ __device__ __forceinline__ uint32_t column_hash(int sx, int sy) {
    uint32_t h = (uint32_t)sx * 0x9E3779B1u ^ (uint32_t)sy * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
// bucket of a column: multiply-high range reduction, so the bucket count need not be a power of two
__host__ __device__ __forceinline__ uint32_t bucket_of(uint32_t colh, uint32_t B) {
    return (uint32_t)(((uint64_t)colh * (uint64_t)B) >> 32);
}
__device__ __forceinline__ uint32_t node_slot_hash(uint32_t colh, int sz) {
    uint32_t g = (colh * 0x9E3779B1u) ^ ((uint32_t)sz * 0xC2B2AE3Du);
    g ^= g >> 16; g *= 0x27D4EB2Fu;
    g ^= g >> 15;
    return g;
}

// points of workgroup w: [w*chunk, min(n, (w+1)*chunk))
__device__ __forceinline__ void wg_range(uint64_t n, uint32_t nwg, uint32_t w, uint64_t& lo, uint64_t& hi) {
    uint64_t chunk = (n + nwg - 1) / nwg;
    chunk = (chunk + 63) & ~63ull;
    lo = (uint64_t)w * chunk;
    hi = lo + chunk;
    if (lo > n) lo = n;
    if (hi > n) hi = n;
}

// one launch that prepares a build: counters, partition flags and the column-first bitmap
__global__ void __launch_bounds__(256) k_part_clear(Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                    uint32_t* __restrict__ bitmap, uint64_t words) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt->num_nodes = 0; cnt->num_columns = 0; cnt->num_slopes = 0; cnt->err_key_range = 0; cnt->err_table_full = 0;
        pc->lds_overflow = 0; pc->stage_overflow = 0;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) bitmap[i] = 0u;
}

// ---------------------------------------------------------------------------------------------
// pass 1: per-workgroup bucket histogram.  hist is [nwg][B].
// ---------------------------------------------------------------------------------------------
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kPartThreads) k_part_hist(const float* __restrict__ xyz, uint64_t n, GridParams P,
                                                            uint32_t B, uint32_t* __restrict__ hist,
                                                            Counters* __restrict__ cnt) {
    extern __shared__ uint32_t lh[];
    for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) lh[i] = 0;
    __syncthreads();
    uint64_t lo, hi;
    wg_range(n, gridDim.x, blockIdx.x, lo, hi);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kPartThreads) {
        const float* p = xyz + i * STRIDE_FLOATS;
        const float px = p[0], py = p[1], pz = p[2];
        PointKey k = point_key(px, py, pz, P.ox, P.oy, P.oz, P.grid_len, P.z_len);
        if (!k.ok) { atomicAdd(&cnt->err_key_range, 1u); continue; }
        atomicAdd(&lh[bucket_of(column_hash(k.sx, k.sy), B)], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + (uint64_t)blockIdx.x * B;
    for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) out[i] = lh[i];
}

// ---------------------------------------------------------------------------------------------
// pass 1b: for every bucket, exclusive scan of its counts over the workgroups (in place) and the
// bucket total.  Block = 32 buckets x 8 workgroup segments.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_part_offsets(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals,
                                                      uint32_t B, uint32_t nwg) {
    __shared__ uint32_t seg[8][32];
    const uint32_t bx = threadIdx.x & 31, wy = threadIdx.x >> 5;
    const uint32_t b = blockIdx.x * 32 + bx;
    const uint32_t per = (nwg + 7) / 8;
    const uint32_t w0 = wy * per, w1 = min(nwg, w0 + per);
    uint32_t s = 0;
    if (b < B)
        for (uint32_t w = w0; w < w1; ++w) s += hist[(uint64_t)w * B + b];
    seg[wy][bx] = s;
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t y = 0; y < wy; ++y) run += seg[y][bx];
    if (b < B) {
        for (uint32_t w = w0; w < w1; ++w) {
            const uint32_t v = hist[(uint64_t)w * B + b];
            hist[(uint64_t)w * B + b] = run;
            run += v;
        }
        if (wy == 7) totals[b] = run;
    }
}

// block-wide exclusive scan of B (<= 32768) bucket totals into LDS `cur`; returns nothing, cur[b] = base of b
__device__ __forceinline__ void block_scan_totals(const uint32_t* __restrict__ totals, uint32_t B, uint32_t* cur,
                                                  uint32_t* wave_sums /*[8]*/) {
    // kPartThreads threads, each owns a contiguous run of `per` buckets
    const uint32_t per = (B + kPartThreads - 1) / kPartThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (b0 + j < B) s += totals[b0 + j];
    // inclusive scan across the wave
    uint32_t incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_sums[w];
    uint32_t run = wbase + incl - s;
    for (uint32_t j = 0; j < per; ++j)
        if (b0 + j < B) { cur[b0 + j] = run; run += totals[b0 + j]; }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// pass 2: scatter {x,y,z,idx} records into bucket order.  bucket_base[B+1] is written by block 0.
// ---------------------------------------------------------------------------------------------
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kPartThreads) k_part_scatter(const float* __restrict__ xyz, uint64_t n, uint32_t first_base,
                                                               GridParams P, uint32_t B,
                                                               const uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ totals,
                                                               uint32_t* __restrict__ bucket_base,
                                                               float4* __restrict__ recs) {
    extern __shared__ uint32_t cur[];
    __shared__ uint32_t wave_sums[kPartThreads / 64];
    block_scan_totals(totals, B, cur, wave_sums);
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) bucket_base[i] = cur[i];
        if (threadIdx.x == 0) bucket_base[B] = cur[B - 1] + totals[B - 1];
    }
    const uint32_t* mine = hist + (uint64_t)blockIdx.x * B;
    for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) cur[i] += mine[i];
    __syncthreads();
    uint64_t lo, hi;
    wg_range(n, gridDim.x, blockIdx.x, lo, hi);
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kPartThreads) {
        const float* p = xyz + i * STRIDE_FLOATS;
        const float px = p[0], py = p[1], pz = p[2];
        PointKey k = point_key(px, py, pz, P.ox, P.oy, P.oz, P.grid_len, P.z_len);
        if (!k.ok) continue;
        const uint32_t pos = atomicAdd(&cur[bucket_of(column_hash(k.sx, k.sy), B)], 1u);
        recs[pos] = make_float4(px, py, pz, __uint_as_float(first_base + (uint32_t)i));
    }
}

// ---------------------------------------------------------------------------------------------
// pass 3: one workgroup per bucket
// ---------------------------------------------------------------------------------------------
template <int kBucketSlots>
struct BucketLds {
    unsigned long long key[kBucketSlots];
    double sum[9][kBucketSlots];
    uint32_t cnt[kBucketSlots];
    uint32_t first[kBucketSlots];
    float mean_z[kBucketSlots];
    uint32_t flags[kBucketSlots];          // bits 0..2 GNDT_FLAG_*, bits 8.. column slot
    unsigned long long ckey[kBucketSlots]; // column table
    uint32_t cfirst[kBucketSlots];
    uint32_t ccnt[kBucketSlots];
    unsigned long long okey[kBucketSlots]; // sort list: (column first << 32) | node first
    uint32_t oslot[kBucketSlots];
    uint32_t n_nodes, n_list, stage_base, overflow;
};

// LDS open-addressing helpers.  They take the __shared__ arrays by reference to their element type's
// address space (template on the array) so that the compiler keeps ds_* instructions; a generic
// `volatile T*` parameter makes it fall back to flat_* loads.  A stale non-empty key cannot exist (keys
// are written once), and a stale EMPTY is settled by the compare-and-swap.
template <int kBucketSlots, typename KeyArray>
__device__ __forceinline__ uint32_t lds_find_or_insert(KeyArray& keys, uint32_t start, uint64_t key, uint32_t* n_new) {
    uint32_t slot = start & (kBucketSlots - 1);
    for (int probe = 0; probe < kBucketSlots; ++probe) {
        const unsigned long long k = keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) {
            const unsigned long long old = atomicCAS(&keys[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) { if (n_new) atomicAdd(n_new, 1u); return slot; }
            if (old == key) return slot;
        }
        slot = (slot + 1) & (kBucketSlots - 1);
    }
    return kBucketSlots;
}
template <int kBucketSlots, typename KeyArray>
__device__ __forceinline__ uint32_t lds_find(const KeyArray& keys, uint32_t start, uint64_t key) {
    uint32_t slot = start & (kBucketSlots - 1);
    for (int probe = 0; probe < kBucketSlots; ++probe) {
        const unsigned long long k = keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) return kBucketSlots;
        slot = (slot + 1) & (kBucketSlots - 1);
    }
    return kBucketSlots;
}

template <int kBucketThreads, int kBucketSlots>
__global__ void __launch_bounds__(kBucketThreads) k_bucket_build(const float4* __restrict__ recs,
                                                                 const uint32_t* __restrict__ bucket_base, GridParams P,
                                                                 StageRow* __restrict__ stage, uint32_t stage_cap,
                                                                 uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                                 uint32_t* __restrict__ ord_ncol, uint32_t* __restrict__ bitmap,
                                                                 Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                                 unsigned long long* __restrict__ dbg) {
    __shared__ BucketLds<kBucketSlots> L;
    constexpr int kBucketFill = (kBucketSlots * 25) / 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // diagnostic phase stamps (shader clock), only when the host passes a buffer: [bucket][8]
#define GNDT_STAMP(k) do { if (dbg && tid == 0) dbg[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    GNDT_STAMP(0);
    // ---- P0: clear ----
    for (int s = tid; s < kBucketSlots; s += kBucketThreads) {
        L.key[s] = kEmptyKey;
        L.ckey[s] = kEmptyKey;
#pragma unroll
        for (int j = 0; j < 9; ++j) L.sum[j][s] = 0.0;
        L.cnt[s] = 0; L.first[s] = 0xFFFFFFFFu;
        L.cfirst[s] = 0xFFFFFFFFu; L.ccnt[s] = 0;
    }
    if (tid == 0) { L.n_nodes = 0; L.n_list = 0; L.stage_base = 0; L.overflow = 0; }
    __syncthreads();

    GNDT_STAMP(1);
    // ---- P1: accumulate this bucket's records into the LDS table ----
    const uint32_t lo = bucket_base[blockIdx.x], hi = bucket_base[blockIdx.x + 1];
    const uint32_t span = hi - lo;
    const uint32_t span_round = (span + 63u) & ~63u;
    // software pipeline: the next record is in flight while this one goes through the LDS atomics
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((uint32_t)tid < span) nxt = recs[lo + tid];
    for (uint32_t off = tid; off < span_round; off += kBucketThreads) {
        const bool live = off < span;
        const float4 r = nxt;
        if (off + kBucketThreads < span) nxt = recs[lo + off + kBucketThreads];
        PointKey k = point_key(r.x, r.y, r.z, P.ox, P.oy, P.oz, P.grid_len, P.z_len);
        const uint64_t key = live ? pack_key(k.sx, k.sy, k.sz) : kEmptyKey;
        double v0 = 0, v1 = 0, v2 = 0;
        if (live) {
            v0 = (double)r.x - axis_centre(k.sx, P.ox, P.grid_len);
            v1 = (double)r.y - axis_centre(k.sy, P.oy, P.grid_len);
            v2 = (double)r.z - axis_centre(k.sz, P.oz, P.z_len);
        }
        double q[9] = {v0, v1, v2, v0 * v0, v0 * v1, v0 * v2, v1 * v1, v1 * v2, v2 * v2};
        uint32_t pidx = __float_as_uint(r.w);
        const uint32_t h = node_slot_hash(column_hash(k.sx, k.sy), k.sz);
        const uint64_t key0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(key >> 32)) << 32) |
                              (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)key);
        const bool uniform = __all(live && key == key0);
        if (uniform) {
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] = wave_sum(q[j]);
            for (int o = 32; o > 0; o >>= 1) pidx = min(pidx, (uint32_t)__shfl_down((int)pidx, o, 64));
            if (lane == 0) {
                const uint32_t s = lds_find_or_insert<kBucketSlots>(L.key, h, key, &L.n_nodes);
                if (s >= kBucketSlots) L.overflow = 1;
                else {
#pragma unroll
                    for (int j = 0; j < 9; ++j) atomicAdd(&L.sum[j][s], q[j]);
                    atomicAdd(&L.cnt[s], 64u);
                    atomicMin(&L.first[s], pidx);
                }
            }
        } else if (live) {
            const uint32_t s = lds_find_or_insert<kBucketSlots>(L.key, h, key, &L.n_nodes);
            if (s >= kBucketSlots) L.overflow = 1;
            else {
#pragma unroll
                for (int j = 0; j < 9; ++j) atomicAdd(&L.sum[j][s], q[j]);
                atomicAdd(&L.cnt[s], 1u);
                atomicMin(&L.first[s], pidx);
            }
        }
    }
    __syncthreads();
    if (L.overflow || L.n_nodes > (uint32_t)kBucketFill) {   // uniform across the block
        if (tid == 0) atomicAdd(&pc->lds_overflow, 1u);
        return;
    }

    GNDT_STAMP(2);
    // Reserve this bucket's staging rows now (one memory-side atomic per bucket): its ~microsecond round trip
    // hides behind P2..P4; the value is only parked in LDS right before the barrier that precedes its use.
    uint32_t stage_base_reg = 0;
    if (tid == kBucketThreads - 1) stage_base_reg = atomicAdd(&cnt->num_nodes, L.n_nodes);
    // ---- P2: per node: fp32 mean-z for the slope test, column registration ----
    for (int s = tid; s < kBucketSlots; s += kBucketThreads) {
        const uint64_t key = L.key[s];
        if (key == kEmptyKey) continue;
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        const uint32_t n = L.cnt[s];
        uint32_t fl = 0;
        float mz = 0.f;
        if (n >= (uint32_t)P.min_points) {
            mz = node_mean_z(n, L.sum[2][s], axis_centre(sz, P.oz, P.z_len));
            fl = 1u;
        }
        L.mean_z[s] = mz;
        const uint32_t cs = lds_find_or_insert<kBucketSlots>(L.ckey, column_hash(sx, sy) * 0x85EBCA77u >> 12, column_key(key), nullptr);
        // the column table has as many slots as the node table and at most as many entries: never full
        atomicMin(&L.cfirst[cs], L.first[s]);
        atomicAdd(&L.ccnt[cs], 1u);
        L.flags[s] = fl | (cs << 8);
    }
    __syncthreads();

    GNDT_STAMP(3);
    // ---- P3: slope labels (OcNode::isSlope, map2D.h:66-108) + sort list ----
    uint32_t my_slopes = 0;
    for (int s = tid; s < kBucketSlots; s += kBucketThreads) {
        const uint64_t key = L.key[s];
        if (key == kEmptyKey) continue;
        uint32_t fl = L.flags[s];
        const uint32_t cs = fl >> 8;
        const uint32_t my_first = L.first[s];
        if (fl & 1u) {
            bool slope = true, down = false;
            if (P.demand == 0) {
                int sx, sy, sz;
                unpack_key(key, sx, sy, sz);
                const uint32_t ch = column_hash(sx, sy);
                const float cz = L.mean_z[s];
                bool up = false;
                int za = level_above(sz), zb = level_below(sz);
                uint32_t t = lds_find<kBucketSlots>(L.key, node_slot_hash(ch, za), pack_key(sx, sy, za));
                if (t < kBucketSlots) {
                    const bool visited = L.first[t] < my_first && (L.flags[t] & 1u);
                    const float oz = visited ? L.mean_z[t] : 0.f;
                    if (fabsf(oz - cz) > P.slope_interval) up = true;
                }
                t = lds_find<kBucketSlots>(L.key, node_slot_hash(ch, zb), pack_key(sx, sy, zb));
                if (t < kBucketSlots) {
                    const bool visited = L.first[t] < my_first && (L.flags[t] & 1u);
                    const float oz = visited ? L.mean_z[t] : 0.f;
                    if (fabsf(oz - cz) > P.slope_interval) down = true;
                }
                slope = !up;
            }
            if (slope) { fl |= 2u; if (down) fl |= 4u; ++my_slopes; }
        }
        const uint32_t pos = atomicAdd(&L.n_list, 1u);
        L.okey[pos] = ((uint64_t)L.cfirst[cs] << 32) | (uint64_t)my_first;
        L.oslot[pos] = (uint32_t)s | ((fl & 7u) << 16);   // flags bits 0..2 ride along (slot < 1024)
    }
    __syncthreads();
    // NOTE: L.flags of OTHER slots is read above (bit 0 only) while this loop rewrites nothing in it.

    GNDT_STAMP(4);
    // ---- P4: reserve the staging rows (one memory-side atomic per bucket, issued early so that its
    //          round trip hides behind the ranking loop) ----
    const uint32_t M = L.n_list;          // == L.n_nodes: every occupied slot is listed

    // ---- P5: rank by counting instead of sorting.  Keys (column first-seen, node first-seen) are
    //          unique, so   rank = #{keys below mine}   is the node's row in in-bucket reference order and
    //          idx_in_col = #{keys below mine in my column}.  Every lane reads the same okey[j]: LDS broadcast.
    uint32_t my_cols = 0;
    uint32_t rank[(kBucketSlots + kBucketThreads - 1) / kBucketThreads];
    uint32_t icol[(kBucketSlots + kBucketThreads - 1) / kBucketThreads];
    {
        int it = 0;
        for (uint32_t i = tid; i < M; i += kBucketThreads, ++it) {
            const unsigned long long mine = L.okey[i];
            const uint32_t cf = (uint32_t)(mine >> 32);
            uint32_t r = 0, c = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < M; ++j) {
                const unsigned long long o = L.okey[j];
                const bool below = o < mine;
                r += below ? 1u : 0u;
                c += (below && (uint32_t)(o >> 32) == cf) ? 1u : 0u;
            }
            rank[it] = r; icol[it] = c;
        }
    }
    if (tid == kBucketThreads - 1) L.stage_base = stage_base_reg;
    __syncthreads();
    GNDT_STAMP(5);
    const uint32_t base = L.stage_base;
    if (base + M > stage_cap) {               // uniform
        if (tid == 0) atomicAdd(&pc->stage_overflow, M);
        return;
    }
    {
        int it = 0;
        for (uint32_t i = tid; i < M; i += kBucketThreads, ++it) {
            const uint32_t packed = L.oslot[i];
            const uint32_t s = packed & 0xFFFFu, fl = (packed >> 16) & 7u;
            const uint64_t key = L.key[s];
            const uint32_t cf = (uint32_t)(L.okey[i] >> 32);
            const uint32_t idx_in_col = icol[it];
            const uint32_t cs = L.flags[s] >> 8;
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
            row.col_first = cf; row.idx_in_col = idx_in_col; row.ncol = L.ccnt[cs];
            for (int k = 0; k < 8; ++k) row.pad[k] = 0;
            const uint32_t dst = base + rank[it];
            stage[dst] = row;
            ord_cf[dst] = cf;
            ord_idx[dst] = idx_in_col;
            if (idx_in_col == 0) {
                ord_ncol[dst] = row.ncol;
                atomicOr(&bitmap[cf >> 5], 1u << (cf & 31u));
                ++my_cols;
            }
        }
    }
    if (my_slopes) atomicAdd(&cnt->num_slopes, my_slopes);
    if (my_cols) atomicAdd(&cnt->num_columns, my_cols);
    __syncthreads();
    GNDT_STAMP(6);
#undef GNDT_STAMP
}

// ---------------------------------------------------------------------------------------------
// two-level exclusive scan of u32 values (optionally popcounts), length read from device memory
// ---------------------------------------------------------------------------------------------
template <bool POPC>
__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(const uint32_t* __restrict__ in, const uint32_t* n_ptr,
                                                              uint32_t n_fixed, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t ws[kScanThreads / 64];
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t start = blockIdx.x * kScanChunk;
    uint32_t s = 0;
    for (uint32_t i = start + threadIdx.x; i < min(n, start + kScanChunk); i += kScanThreads) {
        const uint32_t v = in[i];
        s += POPC ? (uint32_t)__popc(v) : v;
    }
    for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_down((int)s, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += ws[w];
        block_sums[blockIdx.x] = t;
    }
}

template <bool POPC>
__global__ void __launch_bounds__(kScanThreads) k_scan_apply(const uint32_t* __restrict__ in, const uint32_t* n_ptr,
                                                             uint32_t n_fixed, const uint32_t* __restrict__ block_sums,
                                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t ws[kScanThreads / 64];
    __shared__ uint32_t s_base;
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t start = blockIdx.x * kScanChunk;
    if (start >= n) return;
    // base = sum of the block sums before this block (a few hundred at most)
    uint32_t b = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kScanThreads) b += block_sums[i];
    for (int off = 32; off > 0; off >>= 1) b += (uint32_t)__shfl_down((int)b, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += ws[w];
        s_base = t;
    }
    __syncthreads();
    // each thread owns kScanChunk / kScanThreads = 8 consecutive elements
    constexpr int PER = kScanChunk / kScanThreads;
    uint32_t v[PER];
    uint32_t s = 0;
    const uint32_t i0 = start + threadIdx.x * PER;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        uint32_t x = (i0 + j < n) ? in[i0 + j] : 0u;
        v[j] = POPC ? (uint32_t)__popc(x) : x;
        s += v[j];
    }
    uint32_t incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += ws[w];
    uint32_t run = s_base + wbase + incl - s;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
    }
}

// ---------------------------------------------------------------------------------------------
// ordering
// ---------------------------------------------------------------------------------------------
// column rank of every staged node = number of column-first bits below its column's first index
__global__ void __launch_bounds__(kBlock) k_order_rank(const uint32_t* __restrict__ ord_cf, const uint32_t* __restrict__ ord_idx,
                                                       const uint32_t* __restrict__ ord_ncol,
                                                       const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_prefix,
                                                       uint32_t* __restrict__ col_rank, uint32_t* __restrict__ col_size,
                                                       const Counters* __restrict__ cnt, const PartCounters* __restrict__ pc) {
    if (pc->lds_overflow | pc->stage_overflow) return;   // the host re-runs the build; staged rows are incomplete
    const uint32_t n = cnt->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t cf = ord_cf[i];
        const uint32_t w = cf >> 5, bit = cf & 31u;
        const uint32_t rank = word_prefix[w] + (uint32_t)__popc(bitmap[w] & ((1u << bit) - 1u));
        col_rank[i] = rank;
        if (ord_idx[i] == 0) col_size[rank] = ord_ncol[i];
    }
}

__global__ void __launch_bounds__(kBlock) k_order_dest(const uint32_t* __restrict__ col_rank, const uint32_t* __restrict__ ord_idx,
                                                       const uint32_t* __restrict__ col_base, uint32_t* __restrict__ inv,
                                                       const Counters* __restrict__ cnt, const PartCounters* __restrict__ pc) {
    if (pc->lds_overflow | pc->stage_overflow) return;
    const uint32_t n = cnt->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        inv[col_base[col_rank[i]] + ord_idx[i]] = i;
}

// staging rows -> SoA result rows in reference order
__global__ void __launch_bounds__(kBlock) k_emit_rows(const StageRow* __restrict__ stage, const uint32_t* __restrict__ inv,
                                                      OutView out, const Counters* __restrict__ cnt,
                                                      const PartCounters* __restrict__ pc) {
    if (pc->lds_overflow | pc->stage_overflow) return;
    const uint32_t n = cnt->num_nodes;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const StageRow row = stage[inv[r]];
        out.sx[r] = row.sx; out.sy[r] = row.sy; out.sz[r] = row.sz;
        out.count[r] = row.count; out.first_idx[r] = row.first; out.flags[r] = row.flags;
        float rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
        if (row.flags & 1u) node_rough_normal(row.scatter, rough, normal);
        for (int k = 0; k < 3; ++k) { out.mean[3 * r + k] = row.mean[k]; out.normal[3 * r + k] = normal[k]; }
        for (int k = 0; k < 6; ++k) out.cov[6 * r + k] = (float)row.scatter[k];
        out.rough[r] = rough;
    }
}

}  // namespace gndt
