// gndt_partition.hpp — strategy PARTITION: the whole build with every per-node atomic kept in LDS.
//
// The atomic path (gndt_kernels.hpp) spends its time in ~11 memory-side atomics per point (MI355X
// executes device-scope atomics at the memory side, ~20 G requests/s chip-wide).  Here the points are
// first partitioned by COLUMN hash into B buckets, so that one workgroup owns every node of its bucket's
// columns and can keep their statistics, the slope labels and the in-bucket ordering in LDS:
//
//   partition, large builds: two levels through LDS tile sorts, no counting passes (see "Two-level partition" below)
//   k_part2_level1   points -> {x,y,z,idx} records grouped by coarse region        reads 12, writes 16 B/pt
//   k_part2_layout   the buckets' regions from level 1's 1-in-64 sample
//   k_part2_level2   coarse regions -> buckets                                     reads 16, writes 16 B/pt
//   partition, small builds and the fallback: exact single-level counting partition
//   k_part_hist      points -> bucket histogram per workgroup                      reads 12 B/pt
//   k_part_offsets   per-bucket exclusive scan over workgroups (+ bucket totals)
//   k_part_scatter   points -> {x,y,z,idx} records grouped by bucket               reads 12, writes 16 B/pt
//   k_bucket_direct  (gndt_bucket3.hpp) one workgroup per bucket: LDS node table, fp64 LDS atomics, column arrays,
//                    slope labels, mean + fp64 scatter -> 96-B staging rows         reads 16 B/pt, writes 96 + 8 B/node
//   k_scan_*         prefix of the per-word column weights (ColumnOrder)
//   k_order_*        destination row of every node (reference order), inverse permutation
//   k_emit_rows      staging rows -> SoA result in reference order                 reads 96 + 4, writes 80 B/node
//
// Reference semantics are the ones of gndt_kernels.hpp (same gndt_math.hpp arithmetic); only the data
// movement differs.  Anything that does not fit (LDS table overflow, staging overflow, a partition region) raises
// a flag and the host re-runs the build with more room, in the end on the atomic path: results never depend on
// the strategy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gndt_kernels.hpp"

namespace gndt {

constexpr int kPartThreads = 1024;   // k_part_hist / k_part_scatter
// k_bucket_direct (gndt_bucket3.hpp) is a template on <threads, LDS node-table slots>; a bucket holding more distinct nodes
// than slots overflows (the host then re-runs the build with larger tables, in the end on the atomic path).
constexpr int kScanChunk = 2048;     // elements per block in the two-level scans
constexpr int kScanThreads = 256;

struct PartCounters {
    uint32_t lds_overflow;     // buckets whose node table overflowed
    uint32_t stage_overflow;   // nodes that did not fit the staging rows
    uint32_t index_overflow;   // column-first indices beyond the bitmap (table path: stream ran past max_points_hint)
    uint32_t part_overflow;    // two-level partition: reservations beyond a region's fixed capacity
    uint32_t max_fill1;        // two-level partition: fullest level-1 region (true count, also beyond its capacity: the
    uint32_t fp_clashes;       //   host sizes the retry from it); buckets done a second time because a fingerprint named the
    uint32_t l1_ticket;        //   wrong node (gndt_bucket3.hpp; diagnostic); level-1 workgroups that are done (the last one lays
    uint32_t small_fallback;   //   out the buckets' regions; 0 between kernels); k_small_finalize met more nodes than it has threads
    uint32_t pairs;            // bucket kernel: lanes whose two ADJACENT records fell into one node (the cloud's locality inside its buckets)
    uint32_t capture_id;       // HOST MIRROR ONLY: which recorded call produced what the mirrors hold (0: a call the host launched itself).
                               //   Stored by the kernel that sends the counters home; the host compares it with what it knows of that
                               //   capture (gndt_sync): a replay it did not see, buffers reallocated since the recording.
    uint32_t blk_miss;         // blocked buckets (gndt_blocked.hpp): records that do not belong to the block their bucket is (the build is re-run hashed)
    uint32_t lds_retry;        // buckets whose 512-slot table overflowed and that wait for the second pass with 1024 slots (the retry list's length)
    uint32_t l1_err;           // level 1 with FoldClear: points beyond the key range (the bucket kernel adds them to Counters::err_key_range, which
                               //   that level-1 launch itself zeroes)
    unsigned long long cols_slopes;   // bucket kernels: {columns (low word), slopes (high word)} of the buckets done so far — ONE memory-side atomic
                               //   per bucket, on THIS line.  They used to be two atomics on Counters' line, where every bucket's row
                               //   reservation (a returning atomic: the bucket waits for its answer) queued behind them: same-line
                               //   atomics are served at ~90 per us, 2 809 buckets x 3 is the bench scene's whole bucket kernel.
                               //   Folded into Counters::num_columns / num_slopes by the kernel that sends the counters home
                               //   (fold_bucket_counts); bucket kernel 120 -> 105 us, profiles/r06_ablation.txt 11.
};

// (block 0, thread 0 of k_emit_rows / k_place_emit_rows, in front of the copy to the host's mirror)
__device__ __forceinline__ void fold_bucket_counts(Counters* cnt, PartCounters* pc) {
    const unsigned long long cs = pc->cols_slopes;
    if (cs) { cnt->num_columns += (uint32_t)cs; cnt->num_slopes += (uint32_t)(cs >> 32); pc->cols_slopes = 0ull; }
}

// What a level-1 launch zeroes on the side when no k_part_clear precedes the build (Part::cursors_alt, gndt_handle.hpp): the cursors and
// partition counters the NEXT build on the handle will take, and THIS build's bitmap, word weights and Counters — none of which a
// level-1 workgroup reads or adds to (its own cursors / counters were zeroed by the build before).  alt_cursors == nullptr: off.
struct FoldClear {
    uint32_t* alt_cursors; uint32_t n_alt;
    PartCounters* alt_pc;
    uint32_t* bitmap; uint32_t* weight; uint64_t words;
};

struct alignas(16) StageRow {   // 96 bytes, gathered whole by k_emit_rows (round 1 padded it to 128: a quarter of the node traffic)
    int32_t sx, sy, sz;
    uint32_t count, first, flags;
    uint32_t col_first, idx_in_col, ncol;
    float mean[3];
    double scatter[6];          // fp64: the eigen-solve runs in k_emit_rows (chip-wide parallelism)
};
static_assert(sizeof(StageRow) == 96, "StageRow layout");

// A staging row of the PARTITION strategies (round 6): the node's additive statistics in cell-local coordinates as the bucket kernel's
// LDS table held them, plus what its column phases found out — nothing derived.  k_emit_rows<true> gathers the record whole (six
// 16-byte loads) and works mean, scatter and eigen-solve out, chip-wide; until round 5 k_bucket_direct did the moments itself, one
// node per thread, in its latency-bound back half.
struct alignas(16) RawNode {
    unsigned long long key;
    uint32_t count, first;
    double sum[9];              // Sum v (3), Sum v v^T upper triangle (6); v = p - centre(node), fp64
    uint32_t info;              // flags (bit 0 has statistics, 1 slope, 2 down) | index in column << 3
    uint32_t ncol;              // nodes of the column
};
static_assert(sizeof(RawNode) == 96, "RawNode layout");

// What the producer of the staging rows records for every column, at the column's first-seen point index cf (the
// row with idx_in_col == 0 does it).  The final row of a node is then
//     (nodes of the columns first seen before cf) + idx_in_col
//   = exclusive prefix of word_weight up to word cf>>5  +  ncol_at of the earlier columns inside that word  + idx_in_col,
// which needs one scan over the bitmap words and no sort, no column ranks, no scan over the columns.
struct ColumnOrder {
    uint32_t* bitmap;        // bit cf
    uint32_t* word_weight;   // [cf >> 5] += nodes of the column
    uint32_t* ncol_at;       // [cf] = nodes of the column (read only where the bit is set)
};
__device__ __forceinline__ void note_column(const ColumnOrder& O, uint32_t cf, uint32_t ncol) {
    atomicOr(&O.bitmap[cf >> 5], 1u << (cf & 31u));
    atomicAdd(&O.word_weight[cf >> 5], ncol);
    O.ncol_at[cf] = ncol;
}

// Hash of a column.  |sx|, |sy| <= 65535, so the biased indices fit 18 bits: THREE full-rate 24-bit multiplies and two
// xor-shifts (round 5: the middle one was a 32-bit multiply — quarter rate on CDNA, the cost of four instructions, paid per point
// in level 1, level 2 and per node in the bucket kernel; the fold in front of it brings the word's top bits down into the 24 the
// multiply reads).  Columns per bucket come out Poisson-distributed (variance / mean 0.79 .. 1.09) on full lattices of 4 M columns
// and on the columns of the bench scenes for 44 .. 32768 buckets, as with the 32-bit multiply (0.73 .. 1.12; checked offline).
__host__ __device__ __forceinline__ uint32_t column_hash(int sx, int sy) {
    const uint32_t a = (uint32_t)(sx + 65536) & 0x3FFFFu, b = (uint32_t)(sy + 65536) & 0x3FFFFu;
    uint32_t h = a * 0x9E3779u + b * 0x85EBCBu;         // (both operands < 2^24: v_mul_u32_u24)
    h ^= h >> 15; h = (h & 0xFFFFFFu) * 0x1B3C6Du;      // (again)
    h ^= h >> 13;
    return h;
}
// bucket of a column: range reduction of the hash's top 24 bits, so the bucket count need not be a power of two
// (B < 2^24; 24 x 24-bit product: two full-rate instructions instead of a quarter-rate 32 x 32 high multiply)
__host__ __device__ __forceinline__ uint32_t bucket_of(uint32_t colh, uint32_t B) {
    return (uint32_t)(((uint64_t)(colh >> 8) * (uint64_t)(B & 0xFFFFFFu)) >> 24);
}
// contiguous form of a signed index (there is no index 0): ... -2, -1, 1, 2 ... -> ... -2, -1, 0, 1 ...
__host__ __device__ __forceinline__ int contiguous_index(int s) { return s > 0 ? s - 1 : s; }
// Bucket of a column.  Hashed (the default), or — GridParams::blk.on — the spatial block the column lies in; a column outside the
// box the block map was laid out for is clamped into it and `miss` is set (the bucket kernel finds the stranger and the build is
// re-run with hashed buckets: gndt_blocked.hpp).
__host__ __device__ __forceinline__ uint32_t column_bucket(int sx, int sy, const GridParams& P, uint32_t B) {
    if (!P.blk.on) return bucket_of(column_hash(sx, sy), B);
    int bx = (contiguous_index(sx) - P.blk.x0) >> P.blk.shx, by = (contiguous_index(sy) - P.blk.y0) >> P.blk.shy;
    bx = bx < 0 ? 0 : (bx >= P.blk.nx ? P.blk.nx - 1 : bx);
    by = by < 0 ? 0 : (by >= P.blk.ny ? P.blk.ny - 1 : by);
    return (uint32_t)(bx * P.blk.ny + by);
}
// Owner rank of a column when a cloud is sharded over W GPUs and every rank builds the columns it owns (gndt_api_dist.hip).
// A SECOND hash of the column, independent of column_hash: the buckets and LDS slots of the owner's local build are chosen
// by column_hash, whose distribution must not be narrowed by the choice of the owner.
__host__ __device__ __forceinline__ uint32_t owner_of(int sx, int sy, uint32_t W) {
    const uint32_t a = (uint32_t)(sx + 65536) & 0x3FFFFu, b = (uint32_t)(sy + 65536) & 0x3FFFFu;
    uint32_t h = a * 0xC2B2AEu + b * 0x27D4EBu;
    h ^= h >> 13; h *= 0x85EBCA77u;
    h ^= h >> 16; h *= 0x9E3779B1u;
    h ^= h >> 15;
    return (uint32_t)(((uint64_t)(h >> 8) * (uint64_t)(W & 0xFFFFFFu)) >> 24);
}
// Locality-aware ownership: blocks of 32 x 32 columns whose owner all ranks agreed on from a sample of everybody's shard
// (gndt_exchange.hpp: most of a scan-ordered shard's points then already sit on their owner and never cross a link); a column
// of a block that is not in the map — never sampled, or too hot to give to ONE rank — falls back to owner_of.
struct OwnerMap {
    const uint32_t* bkey;    // open addressing: block + 1, 0 = empty; nullptr = no map (hash ownership only)
    const uint8_t* bown;     // owner rank of the block in that slot, 0xFF = hash ownership
    uint32_t mask;
    const uint32_t* full;    // != 0 on the device: the block table could not hold every sampled block (k_owner_vote) and is NOT
                             //   used — the same on every rank, since they all vote the same samples into tables of one size
};
__host__ __device__ __forceinline__ uint32_t owner_block(int sx, int sy) {      // 12 + 12 bits
    return ((((uint32_t)(sx + 65536) & 0x3FFFFu) >> 5) << 12) | (((uint32_t)(sy + 65536) & 0x3FFFFu) >> 5);
}
__host__ __device__ __forceinline__ uint32_t owner_block_slot(uint32_t key) { key *= 0x9E3779B1u; return key ^ (key >> 15); }
__device__ __forceinline__ uint32_t owner_lookup(const OwnerMap& M, int sx, int sy, uint32_t W) {
    if (M.bkey && !(M.full && *M.full)) {
        const uint32_t key = owner_block(sx, sy) + 1u;
        uint32_t s = owner_block_slot(key) & M.mask;
        for (uint32_t probe = 0; probe <= M.mask; ++probe) {
            const uint32_t k = M.bkey[s];
            if (k == key) { const uint32_t o = M.bown[s]; if (o < W) return o; break; }
            if (k == 0u) break;
            s = (s + 1u) & M.mask;
        }
    }
    return owner_of(sx, sy, W);
}
// k_part_hist / k_part_scatter modes
constexpr uint32_t kPartModeOwner = 1u;      // digit = owner of the column among B ranks instead of bucket_of(column_hash, B)
constexpr uint32_t kPartModeRecords = 2u;    // input is 16-B records {x, y, z, index word}: the index word is taken as it is
__device__ __forceinline__ uint32_t part_digit(int sx, int sy, uint32_t B, uint32_t mode, const OwnerMap& M) {
    return (mode & kPartModeOwner) ? owner_lookup(M, sx, sy, B) : bucket_of(column_hash(sx, sy), B);      // (the counting partition: hashed buckets only)
}
__device__ __forceinline__ uint32_t node_slot_hash(uint32_t colh, int sz) {
    uint32_t g = (colh * 0x9E3779B1u) ^ ((uint32_t)sz * 0xC2B2AE3Du);
    g ^= g >> 16; g *= 0x27D4EB2Fu;
    g ^= g >> 15;
    return g;
}

// The column part of a point's key, divide-free (axis_index_fast): what the partition passes need.  The z level is only
// computed where the node is (the bucket kernel), which also reports a z index beyond the key range.
__device__ __forceinline__ void column_of_point(float px, float py, const GridParams& P, int& sx, int& sy, bool& ok) {
    bool und = false;
    float cx = axis_ceil_try(px, P.ox, P.inv_grid, und);
    float cy = axis_ceil_try(py, P.oy, P.inv_grid, und);
    if (und) {                                   // (rare: one branch for both axes)
        cx = ceilf(fabsf(px - P.ox) / P.grid_len);
        cy = ceilf(fabsf(py - P.oy) / P.grid_len);
    }
    ok = (cx <= (float)kMaxXY) && (cy <= (float)kMaxXY);      // (NaN: false)
    // axis_from_ceil, spelled for the instruction count (these kernels issue vector instructions half of their time): clamp by one
    // median-of-three (0 -> 1, beyond the key range -> the limit), the sign applied in fp32
    const float ccx = __builtin_amdgcn_fmed3f(cx, 1.0f, (float)kMaxXY), ccy = __builtin_amdgcn_fmed3f(cy, 1.0f, (float)kMaxXY);
    sx = (int)(px > P.ox ? ccx : -ccx);
    sy = (int)(py > P.oy ? ccy : -ccy);
}

// (a record's index word — weight flags, record_weight, record_index —: gndt_kernels.hpp, next to k_accumulate, which reads records too)

// true (wave-uniformly) iff all 64 lanes are `use` and hold bit-identical coordinates
__device__ __forceinline__ bool wave_all_identical(float px, float py, float pz, bool use) {
    const uint32_t ux = __float_as_uint(px), uy = __float_as_uint(py), uz = __float_as_uint(pz);
    const uint32_t fx = (uint32_t)__builtin_amdgcn_readfirstlane((int)ux), fy = (uint32_t)__builtin_amdgcn_readfirstlane((int)uy),
                   fz = (uint32_t)__builtin_amdgcn_readfirstlane((int)uz);
    return __all(use && ux == fx && uy == fy && uz == fz) != 0;
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

// ---------------------------------------------------------------------------------------------
// How many nodes will this cloud have?  A FIRST build without a hint used to guess n / 4, and a cloud of a few points per node
// (10 M points at z = 0.1 m: 3.07 M nodes) ran two or three times until the estimate had been doubled often enough (5.6 ms for a
// build that takes 0.75).  One pass over the cloud settles it: a HyperLogLog sketch of the node keys — 2^14
// registers per workgroup in LDS (max over the keys of "leading zeros of the hash + 1", by register), merged with one memory-side
// atomicMax per non-empty register and workgroup; the host turns the registers into the estimate (standard error 0.8 %).
// Reads 12 B per point, ~40 instructions per point: 30-60 us for 10 M points, against the re-run it saves.  Only fresh handles
// without a hint pay for it (the node count of a handle's last build is a better estimate than any sketch).
// ---------------------------------------------------------------------------------------------
constexpr int kSketchBits = 14, kSketchRegs = 1 << kSketchBits, kSketchThreads = 1024;
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kSketchThreads) k_node_sketch(const float* __restrict__ xyz, uint64_t n, GridParams P, uint32_t* __restrict__ regs) {
    __shared__ uint32_t lr[kSketchRegs];
    for (int i = threadIdx.x; i < kSketchRegs; i += kSketchThreads) lr[i] = 0u;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * kSketchThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kSketchThreads) {
        const float* p = xyz + i * STRIDE_FLOATS;
        const PointKey k = point_key_fast(p[0], p[1], p[2], P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
        if (!k.ok) continue;
        const uint64_t hsh = mix64(pack_key(k.sx, k.sy, k.sz));
        const uint32_t r = (uint32_t)(hsh >> (64 - kSketchBits));
        const uint32_t rho = (uint32_t)__clzll((long long)((hsh << kSketchBits) | (1ull << (kSketchBits - 1)))) + 1u;
        if (lr[r] < rho) atomicMax(&lr[r], rho);          // (the plain read first: most points do not raise their register)
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSketchRegs; i += kSketchThreads) { const uint32_t v = lr[i]; if (v) atomicMax(&regs[i], v); }
}

// one launch that prepares a build: counters, partition flags and the column-first bitmap
static __global__ void __launch_bounds__(256) k_part_clear(Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                    uint32_t* __restrict__ bitmap, uint32_t* __restrict__ word_weight,
                                                    uint64_t words, uint32_t* __restrict__ cursors, uint32_t n_cursors) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt->num_nodes = 0; cnt->num_columns = 0; cnt->num_slopes = 0; cnt->err_key_range = 0; cnt->err_table_full = 0;
        cnt->part_owned = 1u;                        // (num_nodes counts staged rows from here on, not the table's node list)
        pc->lds_overflow = 0; pc->stage_overflow = 0; pc->index_overflow = 0; pc->part_overflow = 0;
        pc->max_fill1 = 0; pc->fp_clashes = 0; pc->l1_ticket = 0; pc->small_fallback = 0; pc->pairs = 0; pc->lds_retry = 0; pc->blk_miss = 0; pc->l1_err = 0; pc->cols_slopes = 0ull;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) { bitmap[i] = 0u; word_weight[i] = 0u; }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cursors; i += gridDim.x * blockDim.x) cursors[i] = 0u;   // two-level partition
}

// ---------------------------------------------------------------------------------------------
// pass 1: per-workgroup bucket histogram.  hist is [nwg][B].
// ---------------------------------------------------------------------------------------------
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kPartThreads) k_part_hist(const float* __restrict__ xyz, uint64_t n, GridParams P,
                                                            uint32_t B, uint32_t* __restrict__ hist,
                                                            Counters* __restrict__ cnt, uint32_t compress, uint32_t mode, OwnerMap M) {
    extern __shared__ uint32_t lh[];
    for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) lh[i] = 0;
    __syncthreads();
    uint64_t lo, hi;
    wg_range(n, gridDim.x, blockIdx.x, lo, hi);
    const uint64_t hi_round = lo + ((hi - lo + 63) & ~63ull);       // whole waves, so that the wave vote below is defined
    for (uint64_t i = lo + threadIdx.x; i < hi_round; i += kPartThreads) {
        const bool live = i < hi;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (live) { const float* p = xyz + i * STRIDE_FLOATS; px = p[0]; py = p[1]; pz = p[2]; }
        int sx, sy;
        bool kok;
        column_of_point(px, py, P, sx, sy, kok);
        if (live && !kok) atomicAdd(&cnt->err_key_range, 1u);
        const bool use = live && kok;
        // 64 consecutive identical points (the (0,0,0) padding of the reference's clouds) become ONE
        // weighted record: counted once here, written once by k_part_scatter
        const bool same = compress && wave_all_identical(px, py, pz, use);
        if (use && (!same || (threadIdx.x & 63) == 0)) atomicAdd(&lh[part_digit(sx, sy, B, mode, M)], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + (uint64_t)blockIdx.x * B;
    for (uint32_t i = threadIdx.x; i < B; i += kPartThreads) out[i] = lh[i];
}

// ---------------------------------------------------------------------------------------------
// pass 1b: for every bucket, exclusive scan of its counts over the workgroups (in place) and the
// bucket total.  Block = 32 buckets x 8 workgroup segments.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_part_offsets(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals,
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
                                                               float4* __restrict__ recs, uint32_t compress, uint32_t mode, OwnerMap M) {
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
    const uint64_t hi_round = lo + ((hi - lo + 63) & ~63ull);
    for (uint64_t i = lo + threadIdx.x; i < hi_round; i += kPartThreads) {
        const bool live = i < hi;
        float px = 0.f, py = 0.f, pz = 0.f;
        uint32_t word = 0u;
        if (live) {
            const float* p = xyz + i * STRIDE_FLOATS; px = p[0]; py = p[1]; pz = p[2];
            if constexpr (STRIDE_FLOATS == 4) { if (mode & kPartModeRecords) word = __float_as_uint(p[3]); }
        }
        int sx, sy;
        bool kok;
        column_of_point(px, py, P, sx, sy, kok);
        const bool use = live && kok;
        const bool same = compress && wave_all_identical(px, py, pz, use);
        if (use && (!same || (threadIdx.x & 63) == 0)) {
            const uint32_t pos = atomicAdd(&cur[part_digit(sx, sy, B, mode, M)], 1u);
            // bit 31 of the index word marks a record that stands for 64 identical points (lane 0 = the first of them);
            // records that come from another rank's split carry their index word (and weight) with them
            const uint32_t idx = (mode & kPartModeRecords) ? word : ((first_base + (uint32_t)i) | (same ? kWeight64Flag : 0u));
            recs[pos] = make_float4(px, py, pz, __uint_as_float(idx));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level partition without counting passes (large builds).
//
// The single-level scatter above writes 16-B records to ~B = thousands of places at once: every store is its own
// memory transaction (~50 G/s on MI355X), lines leave L2 half-written, and a histogram pass over the cloud has to
// come first.  Here a record moves twice, each time through an LDS tile sort with a fan-out of at most 512:
//   level 1  cloud tile (4096 points)   -> F1 coarse buckets (coarse = fine bucket / F2)
//   level 2  coarse-bucket tile          -> its F2 fine buckets
// A tile counts its digits in LDS, reserves space with ONE memory-side atomic per digit and tile, sorts the tile in
// LDS by digit and copies it out so that consecutive lanes write consecutive records (runs of 4096/F records,
// full lines).  Regions have a fixed capacity (cloud-independent layout, no histogram, no offsets pass): level-1
// region c starts at c*cap1, fine bucket b at b*cap2; the fill comes from the cursors.  A region that would
// overflow raises PartCounters::part_overflow and the host re-runs the build on the exact counting path.
// Record order inside a bucket depends on the order in which tiles reserve; nothing downstream depends on it.
// (Round 5 built level 1 WITHOUT reservations — sorted tiles written back in place, level 2 gathering one segment per tile — and
//  took it out again: level 1 gains 6-19 us of 81, level 2 loses 19-33 of 65 to the per-record segment search, whatever its launch
//  shape; profiles/r05_ablation.txt §1.)
// ---------------------------------------------------------------------------------------------
constexpr int kTileThreads = 512;
#ifndef GNDT_TILE_PER1
#define GNDT_TILE_PER1 8
#endif
#ifndef GNDT_L1_WAVES
#define GNDT_L1_WAVES 4
#endif
constexpr int kTilePer1 = GNDT_TILE_PER1;       // level 1: 4096 records per tile (68 KB of LDS); smaller tiles measured slower here
#ifndef GNDT_TILE_PER2
#define GNDT_TILE_PER2 4
#endif
constexpr int kTilePer2 = GNDT_TILE_PER2;       // level 2: 2048 records per tile (34 KB): four tiles resident per CU
constexpr bool kWeight512Ok = kTilePer1 * 64 == 512;   // kWeight512Flag: a wave's share of a level-1 tile is the 512 points a weighted record can stand for
constexpr int kMaxFan = 512;               // fan-out per level: up to 512 x 512 buckets
// TWO-LEVEL partition: level 1's cursors — one per coarse region, a returning memory-side atomic per region and TILE — lie one per 128-byte
// line: every tile of the cloud reserves in all of them, and packed into two or three lines (53 regions on the bench scene, 2 442 tiles)
// the lines' atomic units serve ~130 k lane-operations one after the other.  One call, A B A B: level 1 73.2 | 74.3 -> 71.3 | 70.8 us on the
// bench scene, 0.808 | 0.787 -> 0.762 | 0.755 ms on the 100 M-point terrain.  NOT for the one-level partition of a small frame (a few
// hundred cursors met by ~200 tiles: spread out, every tile's reservation touches a few hundred lines — level 1 10.2 -> 12.0 us) nor for
// level 2's cursors (only met by the ~50 tiles of their own region).  The kernels take the shift as an argument (cshift).
#ifndef GNDT_CURSOR1_SHIFT
#define GNDT_CURSOR1_SHIFT 5
#endif
constexpr int kCursor1Shift = GNDT_CURSOR1_SHIFT;
constexpr int kCursor1Words = kMaxFan << kCursor1Shift;      // words the level-1 cursors take in Part::cursors
constexpr uint32_t kSampleEvery = 64;      // level 1 samples one record in 64 to size the buckets' regions (the hash test below is >> 26)

// FAN = 256 or 512: the fan-out the LDS arrays are sized for (the small variant keeps four level-2 tiles per CU)
template <int PER, int FAN>
struct TileLds {
    using Digit = typename std::conditional<(FAN <= 256), uint8_t, uint16_t>::type;
    float4 rec[kTileThreads * PER];
    Digit digit[kTileThreads * PER];     // digit of every sorted slot
    uint32_t hist[FAN];
    uint32_t scan[FAN + 1];
    uint32_t gbase[FAN];
    uint32_t dbase[FAN], dcap[FAN];      // individually laid out regions (level 2)
    uint32_t wave_tot[kTileThreads / 64];
};

// r[j] / dig[j]: this thread's records and their digits (0xFFFFFFFF = no record).  cursor[d] counts what is reserved in
// the region of digit d.  Regions are either evenly spaced (dbase == nullptr: digit d at out[region0 + d * region_stride],
// `cap` records each) or laid out individually (digit d at out[dbase[d]], dcap[d] records).
template <int PER, int FAN>
__device__ __forceinline__ void tile_partition(TileLds<PER, FAN>& L, const float4 (&r)[PER], const uint32_t (&dig)[PER],
                                               uint32_t nd, uint32_t* __restrict__ cursor, uint32_t cap, uint64_t region0,
                                               uint64_t region_stride, const uint32_t* __restrict__ dbase,
                                               const uint32_t* __restrict__ dcap, float4* __restrict__ out,
                                               PartCounters* __restrict__ pc, int cshift = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the regions' bases / capacities do not depend on the tile: requested first, used by the copy-out
    uint32_t my_dbase = 0, my_room = cap;
    if (dbase && (uint32_t)tid < nd) { my_dbase = dbase[tid]; my_room = dcap[tid]; }
    for (uint32_t d = tid; d < nd; d += kTileThreads) L.hist[d] = 0u;
    __syncthreads();
    uint32_t rank[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) rank[j] = (dig[j] != 0xFFFFFFFFu) ? atomicAdd(&L.hist[dig[j]], 1u) : 0u;
    __syncthreads();
    // reserve space (one memory-side atomic per digit present in the tile) and scan the tile's counts.  The atomic's
    // round trip (~2 us) is not waited for here: its result is only needed by the copy-out, after the tile is sorted.
    uint32_t c = 0, g = 0;
    if ((uint32_t)tid < nd) {
        c = L.hist[tid];
        if (c) g = atomicAdd(&cursor[(size_t)tid << cshift], c);      // (cshift: kCursor1Shift for level 1's cursors, 0 otherwise)
    }
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) L.wave_tot[wave] = incl;
    __syncthreads();
    if ((uint32_t)tid < nd) {
        uint32_t base = incl - c;
        for (int w = 0; w < wave; ++w) base += L.wave_tot[w];
        L.scan[tid] = base;
        if ((uint32_t)tid == nd - 1) L.scan[nd] = base + c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (dig[j] != 0xFFFFFFFFu) {
            const uint32_t pos = L.scan[dig[j]] + rank[j];
            L.rec[pos] = r[j];
            L.digit[pos] = (typename TileLds<PER, FAN>::Digit)dig[j];
        }
    if ((uint32_t)tid < nd) {
        if (c && g + c > my_room) atomicAdd(&pc->part_overflow, 1u);
        L.gbase[tid] = g;
        if (dbase) { L.dbase[tid] = my_dbase; L.dcap[tid] = my_room; }
    }
    __syncthreads();
    const uint32_t total = L.scan[nd];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                // consecutive lanes copy consecutive sorted slots
        const uint32_t j = (uint32_t)k * kTileThreads + tid;
        if (j < total) {
            const uint32_t d = L.digit[j];
            const uint32_t within = L.gbase[d] + (j - L.scan[d]);
            if (dbase) { if (within < L.dcap[d]) out[(uint64_t)L.dbase[d] + within] = L.rec[j]; }
            else if (within < cap) out[region0 + (uint64_t)d * region_stride + within] = L.rec[j];
        }
    }
}

// Region of every bucket from the sample level 1 took: capacity = 2 x the estimate + 2048 records.  (A bucket of c records
// has c/64 +- sqrt(c/64) votes; it overflows if c > 2 x 64 x votes + 2048, which for c = 2000..5000 is 6 sigma or more away
// and never happens below 2048: with 50 000 buckets, 1.6x + 1024 still overflowed a handful per build.)  LiDAR clouds' hot
// columns simply get the room they need.  base = exclusive prefix.  Run by ONE workgroup of kTileThreads threads — the last
// level-1 workgroup to finish (round 4: a launch of its own cost 6 us of a 0.38 ms build); writes lo[] (= base) and cap[].
struct LayoutLds { uint32_t wsum[16]; uint32_t carry; uint32_t last; };
template <int T>
__device__ __forceinline__ void part2_layout(LayoutLds& S, const uint32_t* est2, uint32_t B, uint32_t* __restrict__ lo,
                                             uint32_t* __restrict__ cap, uint64_t rec_capacity, PartCounters* __restrict__ pc) {
    // A thread takes a contiguous run of buckets, kRun at a time with all of their loads in flight at once (one memory
    // round trip per batch instead of one per element: the launch this replaces walked B / 1024 dependent rounds, 6 us).
    constexpr uint32_t kRun = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) S.carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < B; b0 += (uint32_t)T * kRun) {
        const uint32_t mine = b0 + (uint32_t)tid * kRun;
        uint32_t c[kRun], s = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRun; ++j)      // (the votes were added by other workgroups with memory-side atomics: read them there as well)
            c[j] = mine + j < B ? (uint32_t)((uint64_t)__hip_atomic_load(&est2[mine + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * kSampleEvery * 2) + 2048u : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kRun; ++j) s += c[j];
        uint32_t incl = s;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) S.wsum[wave] = incl;
        __syncthreads();
        uint32_t run = S.carry + incl - s;
        for (int w = 0; w < wave; ++w) run += S.wsum[w];
#pragma unroll
        for (uint32_t j = 0; j < kRun; ++j)
            if (mine + j < B) { cap[mine + j] = c[j]; lo[mine + j] = run; run += c[j]; }
        __syncthreads();
        if (tid == T - 1) S.carry = run;
        __syncthreads();
    }
    if (tid == 0 && (uint64_t)S.carry > rec_capacity) atomicAdd(&pc->part_overflow, 1u);
}

// level 1: the cloud -> coarse regions.  One tile per workgroup.  A coarse region can be split into R sub-regions
// with their own cursors (tile t fills sub-region t % R) to spread the reservations of thousands of tiles over
// more words (same-address atomics serialise at the memory side); with 4096-point tiles R = 1 measured best.
// OWNER: the same kernel as the split of a sharded cloud by column owner (gndt_api_dist.hip): the digit is the owner rank of the
// point's column among B ranks (region c at c * cap1, nothing sampled), everything else — the pipelined tile loads, the folding
// of identical points into weighted records, the LDS sort and the coalesced copy-out — is what level 1 does anyway.
// PER1: points per thread and tile.  kTilePer1 (4096-point tiles) everywhere but on small clouds (the one-level partition of < 1 M
// points), where kTilePerSmall makes 1024-point tiles: a 200 k-point frame is 49 tiles of 4096 — a fifth of the CUs busy for 11-14 us —
// or 196 of 1024.  (A wave's share of such a tile is 128 points: no weight-512 records, the weight-64 ones stay.)
constexpr int kTilePerSmall = 2;
template <int STRIDE_FLOATS, int FAN, bool IDXW = false, bool OWNER = false, int PER1 = kTilePer1>
__global__ void __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(GNDT_L1_WAVES, GNDT_L1_WAVES))) k_part2_level1(const float* __restrict__ xyz, uint64_t n, uint32_t first_base,
                                                               GridParams P, uint32_t B, uint32_t F1, uint32_t F2_shift, uint32_t R,
                                                               uint32_t* __restrict__ cursor1, uint32_t cap1,
                                                               uint32_t* __restrict__ est2,
                                                               float4* __restrict__ recs1, Counters* __restrict__ cnt,
                                                               PartCounters* __restrict__ pc, uint32_t compress, OwnerMap M,
                                                               uint32_t* __restrict__ lay_lo, uint32_t* __restrict__ lay_cap, uint64_t rec_capacity,
                                                               FoldClear F, int cshift) {
    constexpr int PER = PER1;
    __shared__ TileLds<PER, FAN> L;
    // Persistent workgroups, software-pipelined: the loads of tile t+1 are in flight while tile t is keyed, sorted
    // in LDS and copied out (the kernel is latency-bound: ~70 % of a wave's life is spent parked on waits).
    const uint64_t ntiles = (n + (uint64_t)kTileThreads * PER - 1) / ((uint64_t)kTileThreads * PER);
    float nx[PER], ny[PER], nz[PER];
    auto load_tile = [&](uint64_t tile) {
        const uint64_t t0 = tile * (kTileThreads * PER);
        const float* __restrict__ base = xyz + t0 * STRIDE_FLOATS;            // uniform: the 64-bit arithmetic stays scalar
        const uint32_t have = (uint32_t)min((uint64_t)(kTileThreads * PER), n - t0);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t i = (uint32_t)(j * kTileThreads) + threadIdx.x;     // a wave holds 64 consecutive points
            nx[j] = 0.f; ny[j] = 0.f; nz[j] = 0.f;
            if (i < have) { const float* p = base + i * (uint32_t)STRIDE_FLOATS; nx[j] = p[0]; ny[j] = p[1]; nz[j] = p[2]; }
        }
    };
    uint64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    const bool fold = F.alt_cursors != nullptr;            // (uniform)
    if (fold) {                                            // (behind the first tile's loads: stores nobody in this launch looks at)
        const uint64_t g = (uint64_t)blockIdx.x * kTileThreads + threadIdx.x, gs = (uint64_t)gridDim.x * kTileThreads;
        for (uint64_t i = g; i < F.words; i += gs) { F.bitmap[i] = 0u; F.weight[i] = 0u; }
        for (uint64_t i = g; i < F.n_alt; i += gs) F.alt_cursors[i] = 0u;
        if (g == 0) {
            *F.alt_pc = PartCounters{};
            cnt->num_nodes = 0; cnt->num_columns = 0; cnt->num_slopes = 0; cnt->err_key_range = 0; cnt->err_table_full = 0;
            cnt->part_owned = 1u;
        }
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * (kTileThreads * PER);
        const uint32_t rep = R > 1u ? (uint32_t)(tile % R) : 0u;
        const bool r_is_one = R <= 1u;                     // (uniform; what the host passes today: no multiply on the per-point chain)
        const uint32_t have_now = (uint32_t)min((uint64_t)(kTileThreads * PER), n - t0);     // points of this tile (32-bit compares below)
        const uint32_t idx0 = first_base + (uint32_t)t0;
        float cx[PER], cy[PER], cz[PER];
        // IDXW: the input is 16-B records whose 4th word is the index word (taken as it is).  It is not prefetched with the
        // coordinates (eight more registers per tile in flight): its line arrived with them and is read when the tile is taken up.
        uint32_t cw[IDXW ? PER : 1];
#pragma unroll
        for (int j = 0; j < PER; ++j) { cx[j] = nx[j]; cy[j] = ny[j]; cz[j] = nz[j]; }
        if constexpr (IDXW) {
            const float* __restrict__ base = xyz + t0 * STRIDE_FLOATS;
            const uint32_t have = (uint32_t)min((uint64_t)(kTileThreads * PER), n - t0);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t i = (uint32_t)(j * kTileThreads) + threadIdx.x;
                cw[j] = i < have ? __float_as_uint(base[i * (uint32_t)STRIDE_FLOATS + 3]) : 0u;
            }
        }
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);
        // All 512 points this wave holds in the tile bit-identical (a stretch of the converters' zero padding)?  Then they go
        // out as ONE record of weight 512 instead of eight of weight 64: the bucket that collects the padding gets 8x fewer.
        bool all8 = kWeight512Ok && PER == kTilePer1 && !IDXW && compress != 0u && t0 + (uint64_t)kTileThreads * PER <= n;
        if constexpr (!IDXW) {
            const uint32_t fx = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cx[0])),
                           fy = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cy[0])),
                           fz = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cz[0]));
            bool mine = true;
#pragma unroll
            for (int j = 0; j < PER; ++j) mine = mine && __float_as_uint(cx[j]) == fx && __float_as_uint(cy[j]) == fy && __float_as_uint(cz[j]) == fz;
            all8 = all8 && __all(mine) != 0;
        }
        float4 r[PER];
        uint32_t dig[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t il = (uint32_t)j * kTileThreads + threadIdx.x;              // (index inside the tile)
            const uint32_t i = idx0 + il;                                              // first-seen index of the point
            const bool live = il < have_now;
            const float px = cx[j], py = cy[j], pz = cz[j];
            int sx, sy;
            bool kok;
            column_of_point(px, py, P, sx, sy, kok);
            if (live && !kok) atomicAdd(fold ? &pc->l1_err : &cnt->err_key_range, 1u);
            const bool use = live && kok;
            const bool same = !IDXW && (all8 || (compress && wave_all_identical(px, py, pz, use)));   // 64 identical points -> one weighted record
            dig[j] = 0xFFFFFFFFu;
            if (use && (!same || ((threadIdx.x & 63) == 0 && (!all8 || j == 0)))) {
                if constexpr (OWNER) {
                    dig[j] = owner_lookup(M, sx, sy, B);
                } else {
                    const uint32_t b = column_bucket(sx, sy, P, B);
                    dig[j] = r_is_one ? (b >> F2_shift) : (b >> F2_shift) * R + rep;  // F2 is a power of two; sub-region rep of coarse region b >> F2_shift
                    // One record in kSampleEvery votes for its bucket (k_part2_layout sizes the buckets' regions from the
                    // votes).  Chosen by a hash of the point index: a fixed stride would alias with the scan pattern of a
                    // spinning LiDAR (the same azimuths every ring) and with the lane-0 records of compressed waves.
                    if ((((i & 0xFFFFFFu) * 0x9E3779u) >> 26) == 0u) atomicAdd(&est2[b], 1u);     // (a full-rate 24-bit multiply; the pattern repeats every 2^24 points)
                }
                uint32_t idx = i | (same ? (all8 ? (kWeight64Flag | kWeight512Flag) : kWeight64Flag) : 0u);
                if constexpr (IDXW) idx = cw[j];                                   // (the host passes compress = 0 with records)
                r[j] = make_float4(px, py, pz, __uint_as_float(idx));
            }
        }
        tile_partition<PER, FAN>(L, r, dig, OWNER ? B : F1 * R, cursor1, cap1, 0ull, (uint64_t)cap1, nullptr, nullptr, recs1, pc, cshift);
        __syncthreads();                                   // the tile's LDS image is reused by the next iteration
    }
    // Two-level partition: the LAST workgroup to get here lays out the buckets' regions for level 2 from everybody's votes
    // (lay_lo == nullptr: one-level partition / owner split / not the last launch over a segmented input: nothing to lay out).
    if constexpr (!OWNER) {
        if (lay_lo) {                                      // uniform
            LayoutLds& S = *reinterpret_cast<LayoutLds*>(&L);      // (the tile image is dead)
            // This workgroup's votes are memory-side atomics: once they are acknowledged (vmcnt) they are where the last workgroup
            // will read them, also memory-side.  NOT __threadfence(): an agent-scope release writes the L2 back — the records this
            // kernel has just written — and cost 0.3 ms per build when every workgroup did it.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) S.last = atomicAdd(&pc->l1_ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
            __syncthreads();
            if (S.last) {
                part2_layout<kTileThreads>(S, est2, B, lay_lo, lay_cap, rec_capacity, pc);
                if (threadIdx.x == 0) pc->l1_ticket = 0u;
            }
        }
    }
}

// level 2: sub-region blockIdx.y (of coarse region blockIdx.y / R), tile blockIdx.x of it -> that region's fine buckets
template <int FAN>
__global__ void __launch_bounds__(kTileThreads) k_part2_level2(const float4* __restrict__ recs1, const uint32_t* __restrict__ cursor1,
                                                               uint32_t cap1, uint32_t R, GridParams P, uint32_t B, uint32_t F2,
                                                               uint32_t* __restrict__ cursor2, const uint32_t* __restrict__ lo,
                                                               const uint32_t* __restrict__ cap, float4* __restrict__ recs2,
                                                               PartCounters* __restrict__ pc) {
    constexpr int PER = kTilePer2;
    __shared__ TileLds<PER, FAN> L;
    const uint32_t v = blockIdx.y, c = v / R;
    // the fullest level-1 region (true count, also beyond its capacity: the host sizes a retry from it)
    const uint32_t fill1 = cursor1[(size_t)v << kCursor1Shift];
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&pc->max_fill1, fill1);
    if (pc->part_overflow) return;                       // level 1 or the layout already gave up: the build is re-run
    const uint32_t have = min(fill1, cap1);
    const uint32_t t0 = blockIdx.x * (kTileThreads * PER);
    if (t0 >= have) return;
    const uint32_t b0 = c * F2, nd = min(F2, B - b0);
    const float4* src = recs1 + (uint64_t)v * cap1;
    float4 r[PER];
    uint32_t dig[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t i = t0 + (uint32_t)j * kTileThreads + threadIdx.x;
        dig[j] = 0xFFFFFFFFu;
        if (i < have) {
            r[j] = src[i];
            int sx, sy;
            bool kok;
            column_of_point(r[j].x, r[j].y, P, sx, sy, kok);
            dig[j] = column_bucket(sx, sy, P, B) - b0;
        }
    }
    tile_partition<PER, FAN>(L, r, dig, nd, cursor2 + b0, 0u, 0ull, 0ull, lo + b0, cap + b0, recs2, pc);
}

// Where bucket b's records are, for the bucket kernel (no kernel of its own: two or three loads per bucket).
//   exact counting partition : [base[b], base[b + 1])                                  (fill == nullptr)
//   two-level partition      : lo[b] + [0, min(fill[b], cap[b]))   lo / cap from k_part2_layout, fill = the level-2 cursors
//   one-level tile partition : b * stride + [0, min(fill[b], stride))                  (lo == nullptr; fill = the level-1 cursors)
struct BucketRanges {
    const uint32_t* lo;
    const uint32_t* cap;
    const uint32_t* fill;
    uint32_t stride;
};
__device__ __forceinline__ void bucket_range(const BucketRanges& R, uint32_t b, uint32_t& lo, uint32_t& hi) {
    if (!R.fill) { lo = R.lo[b]; hi = R.lo[b + 1]; }
    else if (R.lo) { lo = R.lo[b]; hi = lo + min(R.fill[b], R.cap[b]); }
    else { lo = b * R.stride; hi = lo + min(R.fill[b], R.stride); }      // (one-level partition: level 1's cursors — dense there — are the fills)
}

// ---------------------------------------------------------------------------------------------
// pass 3: one workgroup per bucket
// ---------------------------------------------------------------------------------------------
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

// the same prefix in ONE launch of one workgroup, for short arrays (small clouds: two launches cost more than the scan)
constexpr uint32_t kScanSmallMax = 1u << 13;      // (7.7 us at 6 k words and linear in them: 52 us at 31 k, where the two-launch scan takes 10)
static __global__ void __launch_bounds__(1024) k_scan_small(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t ws[16];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t i0 = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t j = 0; j < per; ++j) if (i0 + j < n) s += in[i0 + j];
    uint32_t incl = s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t run = incl - s;
    for (int w = 0; w < wave; ++w) run += ws[w];
    for (uint32_t j = 0; j < per; ++j) if (i0 + j < n) { const uint32_t v = in[i0 + j]; out[i0 + j] = run; run += v; }
}

// ---------------------------------------------------------------------------------------------
// ordering
// ---------------------------------------------------------------------------------------------
// Staging rows grouped by column (k_bucket_direct): a column's rows are adjacent and in first-seen order, its first row's
// ord_idx is kOrdHeadFlag | column size.  One lookup of the column's place, then its rows in a run.
constexpr uint32_t kOrdHeadFlag = 0x80000000u;
// SCAN (small clouds: the index range's bitmap words fit LDS): no scan launch in front of this kernel — every workgroup works the
// prefix of the word weights out for itself, in LDS (word_weight, words), and looks the word bases up there; workgroup 0 leaves
// them in word_base as the scan launches would have.  A 200 k-point frame is seven launches of 4-25 us each, two of them the scan.
constexpr uint32_t kDestScanMax = 1u << 13;
template <bool SCAN>
static __global__ void __launch_bounds__(kBlock) k_order_dest_columns(const uint32_t* __restrict__ ord_cf, const uint32_t* __restrict__ ord_idx,
                                                                      const uint32_t* __restrict__ bitmap, uint32_t* __restrict__ word_base,
                                                                      const uint32_t* __restrict__ ncol_at, uint32_t* __restrict__ inv,
                                                                      const Counters* __restrict__ cnt, const PartCounters* __restrict__ pc,
                                                                      const uint32_t* __restrict__ word_weight, uint32_t words) {
    if (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) return;
    const uint32_t n = cnt->num_nodes;
    __shared__ uint32_t s_base[SCAN ? kDestScanMax : 1];
    __shared__ uint32_t s_wave[kBlock / 64];
    if (SCAN) {
        for (uint32_t j = threadIdx.x; j < words; j += kBlock) s_base[j] = word_weight[j];
        __syncthreads();
        // a thread's words are consecutive, an odd number of them (its neighbours' then start in other banks)
        const uint32_t per = ((words + kBlock - 1u) / kBlock) | 1u;
        const uint32_t j0 = min(threadIdx.x * per, words), j1 = min(j0 + per, words);
        uint32_t sum = 0;
        for (uint32_t j = j0; j < j1; ++j) sum += s_base[j];
        uint32_t incl = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        for (uint32_t j = j0; j < j1; ++j) { const uint32_t v = s_base[j]; s_base[j] = run; run += v; }
        __syncthreads();
        if (blockIdx.x == 0) for (uint32_t j = threadIdx.x; j < words; j += kBlock) word_base[j] = s_base[j];
    }
    // The grid is capped (grid_for), so on a large map a thread walks ~20 rows, and every row of a column head is three DEPENDENT
    // memory round trips (order key, first-seen index, bitmap word + word base).  Round 5: four rows per step, each stage's loads
    // issued for all four before any is used: 0.356 -> 0.329 ms on 100 M points, 0.124 -> 0.110 on 32 M.  (What bounds it then is the
    // number of random accesses — 4.3 M columns over a 3.1 M-word index range; {bitmap word, word base} side by side in one 8-byte
    // record took the pass to 0.309 ms and gave the 0.03 ms back in the scan that has to write the records: not kept.  Folding the
    // pass into the scan — the bitmap walked in order, the gather list written front to back from one {nodes, first row} record
    // per column — is 0.27 ms on that scene and 146 us instead of 30 on the bench scene, whose 160 k columns are all first seen
    // within the first 31 k bitmap words: a handful of threads then place them all.  Not kept either.)
    constexpr int K = 4;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += K * stride) {
        uint32_t v[K], cf[K], bw[K], wb[K];
#pragma unroll
        for (int q = 0; q < K; ++q) { const uint32_t i = i0 + (uint32_t)q * stride; v[q] = i < n ? ord_idx[i] : 0u; }
#pragma unroll
        for (int q = 0; q < K; ++q) cf[q] = (v[q] & kOrdHeadFlag) ? ord_cf[i0 + (uint32_t)q * stride] : 0u;
#pragma unroll
        for (int q = 0; q < K; ++q) { bw[q] = 0u; wb[q] = 0u; if (v[q] & kOrdHeadFlag) { bw[q] = bitmap[cf[q] >> 5]; wb[q] = SCAN ? s_base[cf[q] >> 5] : word_base[cf[q] >> 5]; } }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (!(v[q] & kOrdHeadFlag)) continue;
            const uint32_t i = i0 + (uint32_t)q * stride, nc = v[q] & ~kOrdHeadFlag, w = cf[q] >> 5;
            uint32_t m = bw[q] & ((1u << (cf[q] & 31u)) - 1u);    // columns first seen earlier inside the same word
            uint32_t row = wb[q];
            while (m) { row += ncol_at[(w << 5) + (uint32_t)__builtin_ctz(m)]; m &= m - 1u; }
            for (uint32_t k = 0; k < nc; ++k) inv[row + k] = i + k;
        }
    }
}

// destination row of every staged node (see ColumnOrder), as the inverse permutation the emit kernel gathers by
// `row_of` (nullable): the row of every staged node, kept for the incremental finalisation.  `partial`: only the nodes of columns
// from bitmap word cnt->first_word on are placed again — nothing in front of that word moved in this frame (gndt_table.hpp).
static __global__ void __launch_bounds__(kBlock) k_order_dest(const uint32_t* __restrict__ ord_cf, const uint32_t* __restrict__ ord_idx,
                                                       const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_base,
                                                       const uint32_t* __restrict__ ncol_at, uint32_t* __restrict__ inv,
                                                       const Counters* __restrict__ cnt, const PartCounters* __restrict__ pc,
                                                       uint32_t* __restrict__ row_of, uint32_t partial) {
    if (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) return;   // the host re-runs the build; staged rows are incomplete
    const uint32_t n = cnt->num_nodes;
    const uint32_t w0 = partial ? cnt->first_word : 0u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t cf = ord_cf[i];
        const uint32_t w = cf >> 5;
        if (w < w0) continue;
        uint32_t m = bitmap[w] & ((1u << (cf & 31u)) - 1u);       // columns first seen earlier inside the same word (rare)
        uint32_t row = word_base[w] + ord_idx[i];
        while (m) { row += ncol_at[(w << 5) + (uint32_t)__builtin_ctz(m)]; m &= m - 1u; }
        inv[row] = i;
        if (row_of) row_of[i] = row;
    }
}

// staging rows -> SoA result rows in reference order
// host_cnt / host_pc (nullable): the host's pinned mirrors of the counters and overflow flags.  Both are final when this —
// the last kernel of a PARTITION build — starts, so one lane stores them straight into host memory: two copy commands less
// behind every build (4.5 us each on the stream of a 70 us small-frame build).
__device__ __forceinline__ void emit_one_row(const StageRow& row, uint32_t r, const OutView& out, uint32_t* __restrict__ row_ncol) {
    out.sx[r] = row.sx; out.sy[r] = row.sy; out.sz[r] = row.sz;
    out.count[r] = row.count; out.first_idx[r] = row.first; out.flags[r] = row.flags;
    row_ncol[r] = row.idx_in_col == 0u ? row.ncol : 0u;        // the consumers' column index (gndt_cost.hpp)
    float rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
    if (row.flags & 1u) node_rough_normal(row.scatter, rough, normal);
    for (int k = 0; k < 3; ++k) { out.mean[3 * r + k] = row.mean[k]; out.normal[3 * r + k] = normal[k]; }
    for (int k = 0; k < 6; ++k) out.cov[6 * r + k] = (float)row.scatter[k];
    out.rough[r] = rough;
}

// Small clouds (round 6): the ordering pass and the emit pass as ONE kernel, in scatter form.  Every staged node works its final row out
// itself — its ord_cf names the column's place (bitmap word, word base, the few earlier columns of the word), its record the index in
// the column — reads its record where it lies (consecutive threads, consecutive 96-byte
// records: no gather), does the moments and the eigen-solve and writes its result row where it belongs.  The writes are pieces of 4 to
// 24 bytes at scattered rows: the L2 takes them for a map of a few hundred thousand nodes, and a launch of 9-10 us is gone from a
// 50-60 us build (k_order_dest_columns + k_emit_rows: 9.5 + 9.4 us on the 200 k-point frame; this: see profiles/r06_ablation.txt 6).
// Larger maps keep the two kernels: their result rows are written in whole lines there.
// SCAN: as in k_order_dest_columns — every workgroup works the prefix of the word weights out for itself, in LDS.
template <bool SCAN>
static __global__ void __launch_bounds__(kBlock) k_place_emit_rows(const RawNode* __restrict__ raw, const uint32_t* __restrict__ ord_cf,
                                                                   const uint32_t* __restrict__ ord_idx, const uint32_t* __restrict__ bitmap,
                                                                   uint32_t* __restrict__ word_base, const uint32_t* __restrict__ ncol_at,
                                                                   OutView out, uint32_t* __restrict__ row_ncol,
                                                                   Counters* cnt, PartCounters* pc,
                                                                   Counters* __restrict__ host_cnt, PartCounters* __restrict__ host_pc,
                                                                   uint32_t capture_id, GridParams P,
                                                                   const uint32_t* __restrict__ word_weight, uint32_t words) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        if (threadIdx.x == 0) fold_bucket_counts(cnt, pc);
        if (threadIdx.x == 0 && host_cnt) *host_cnt = *cnt;
        if (threadIdx.x == 1 && host_pc) { *host_pc = *pc; host_pc->capture_id = capture_id; }
    }
    if (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) return;
    const uint32_t n = cnt->num_nodes;
    __shared__ uint32_t s_base[SCAN ? kDestScanMax : 1];
    __shared__ uint32_t s_wave[kBlock / 64];
    if (SCAN) {
        for (uint32_t j = threadIdx.x; j < words; j += kBlock) s_base[j] = word_weight[j];
        __syncthreads();
        const uint32_t per = ((words + kBlock - 1u) / kBlock) | 1u;
        const uint32_t j0 = min(threadIdx.x * per, words), j1 = min(j0 + per, words);
        uint32_t sum = 0;
        for (uint32_t j = j0; j < j1; ++j) sum += s_base[j];
        uint32_t incl = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if (lane >= off) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        for (uint32_t j = j0; j < j1; ++j) { const uint32_t v = s_base[j]; s_base[j] = run; run += v; }
        __syncthreads();
        if (blockIdx.x == 0) for (uint32_t j = threadIdx.x; j < words; j += kBlock) word_base[j] = s_base[j];     // (what the scan launches would have left)
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const RawNode rec = raw[i];                             // (coalesced: staging order)
        const uint32_t icol = rec.info >> 3, fl = rec.info & 7u;
        const uint32_t cf = ord_cf[i];                          // (every row of a column carries the column's first-seen index)
        const uint32_t w = cf >> 5;
        uint32_t m = bitmap[w] & ((1u << (cf & 31u)) - 1u);    // columns first seen earlier inside the same word
        uint32_t r = (SCAN ? s_base[w] : word_base[w]) + icol;
        while (m) { r += ncol_at[(w << 5) + (uint32_t)__builtin_ctz(m)]; m &= m - 1u; }
        int sx, sy, sz;
        unpack_key(rec.key, sx, sy, sz);
        out.sx[r] = sx; out.sy[r] = sy; out.sz[r] = sz;
        out.count[r] = rec.count; out.first_idx[r] = rec.first; out.flags[r] = fl;
        row_ncol[r] = icol == 0u ? rec.ncol : 0u;
        float mean[3] = {0.f, 0.f, 0.f}, rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
        double S[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (fl & 1u) {
            const double c[3] = {axis_centre(sx, P.ox, P.grid_len), axis_centre(sy, P.oy, P.grid_len), axis_centre(sz, P.oz, P.z_len)};
            node_moments(rec.count, rec.sum, c, mean, S);
            node_rough_normal(S, rough, normal);
        }
        out.rough[r] = rough;
#pragma unroll
        for (int k = 0; k < 3; ++k) { out.mean[3 * (size_t)r + k] = mean[k]; out.normal[3 * (size_t)r + k] = normal[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) out.cov[6 * (size_t)r + k] = (float)S[k];
    }
}

// Incremental finalisation (gndt_update*): what k_emit_rows needs to emit only what changed.  Rows in front of
// word_base[first_word] did not move in this frame; of those, the ones whose node a frame touched (work list `touched`, rows
// `row_of`) are written again in place, everything from that row on is gathered as usual.
struct EmitPartial {
    const uint32_t* word_base;   // nullptr: emit every row
    const uint32_t* row_of;
    const uint32_t* touched;
};

// RAW (the PARTITION strategies, round 6): `stage` is an array of RawNode records — mean, scatter and eigen-solve from the node's
// additive statistics here, where the whole chip works on them.
#ifndef GNDT_EMIT_WAVES
#define GNDT_EMIT_WAVES 0
#endif
template <bool RAW = false>
static __global__ void __launch_bounds__(kBlock)
#if GNDT_EMIT_WAVES
__attribute__((amdgpu_waves_per_eu(GNDT_EMIT_WAVES, GNDT_EMIT_WAVES)))
#endif
k_emit_rows(const StageRow* __restrict__ stage, const uint32_t* __restrict__ inv,
                                                      OutView out, uint32_t* __restrict__ row_ncol,
                                                      Counters* cnt, PartCounters* pc,
                                                      Counters* __restrict__ host_cnt, PartCounters* __restrict__ host_pc,
                                                      Counters* tab_cnt, uint32_t advance, EmitPartial part, uint32_t capture_id,
                                                      GridParams P) {
    // (cnt is NOT __restrict__: on the table path tab_cnt points at the same object and lane 0 writes through it below)
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        if (threadIdx.x == 0) fold_bucket_counts(cnt, pc);
        if (threadIdx.x == 0 && host_cnt) *host_cnt = *cnt;
        if (threadIdx.x == 1 && host_pc) { *host_pc = *pc; host_pc->capture_id = capture_id; }
        // Table path (gndt_update*): the end-of-frame bookkeeping rides here as well — how many nodes own a column entry, the
        // next epoch, the stream position of the next frame — instead of two one-thread launches per frame.  None of these
        // fields is read by this kernel or mirrored for the host (which keeps its own stream position); n_work and first_word,
        // which this kernel does read, are reset by the first kernel of the next finalisation.
        if (threadIdx.x == 0 && tab_cnt) {
            tab_cnt->prev_nodes = tab_cnt->num_nodes;
            tab_cnt->n_touched = 0; tab_cnt->n_tcols = 0;
            tab_cnt->epoch = tab_cnt->epoch + 1u;
            tab_cnt->stream_pos += advance;
        }
    }
    if (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) return;
    const uint32_t n = cnt->num_nodes;
    uint32_t r0 = 0;
    if (part.word_base) {
        const uint32_t w0 = cnt->first_word;
        r0 = w0 == 0xFFFFFFFFu ? n : part.word_base[w0];
        const uint32_t nw = cnt->n_work;
        for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nw; j += gridDim.x * blockDim.x) {
            const uint32_t t = part.touched[j], r = part.row_of[t];
            if (r < r0) { const StageRow row = stage[t]; emit_one_row(row, r, out, row_ncol); }          // touched, but where it was
        }
    }
    // A wave takes 64 consecutive rows.  mean / normal ([n][3]) and cov ([n][6]) are written TRANSPOSED through LDS: every store
    // instruction then covers 256 contiguous bytes instead of 64 four-byte pieces 12 or 24 bytes apart (twelve instructions that
    // touched 6-12 lines each: the kernel spent half its time stalled at issue behind them).
    __shared__ float s_t[kBlock / 64][64 * 12];
    const int lane = threadIdx.x & 63;
    float* const wm = &s_t[threadIdx.x >> 6][0];       // [64][3] mean
    float* const wn = wm + 64 * 3;                      // [64][3] normal
    float* const wc = wm + 64 * 6;                      // [64][6] cov
    // (the index of the NEXT iteration's staging row is requested one iteration ahead: on a large map a wave walks ~20 iterations of
    //  two dependent round trips each — index, then row)
    const uint32_t rb_first = r0 + blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
    uint32_t src_next = (rb_first + (uint32_t)lane) < n ? inv[rb_first + (uint32_t)lane] : 0u;
    for (uint32_t rb = rb_first; rb < n; rb += gridDim.x * blockDim.x) {   // (wave-uniform)
        const uint32_t r = rb + (uint32_t)lane;
        const uint32_t src = src_next;
        { const uint32_t rn = r + gridDim.x * blockDim.x; src_next = rn < n ? inv[rn] : 0u; }
        if (r < n) {
            if constexpr (RAW) {
                const RawNode rec = reinterpret_cast<const RawNode*>(stage)[src];      // (the 96-byte record in one piece)
                const uint32_t fl = rec.info & 7u;
                int sx, sy, sz;
                unpack_key(rec.key, sx, sy, sz);
                out.sx[r] = sx; out.sy[r] = sy; out.sz[r] = sz;
                out.count[r] = rec.count; out.first_idx[r] = rec.first; out.flags[r] = fl;
                row_ncol[r] = (rec.info >> 3) == 0u ? rec.ncol : 0u;
                float mean[3] = {0.f, 0.f, 0.f}, rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
                double S[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                if (fl & 1u) {
                    const double c[3] = {axis_centre(sx, P.ox, P.grid_len), axis_centre(sy, P.oy, P.grid_len), axis_centre(sz, P.oz, P.z_len)};
                    node_moments(rec.count, rec.sum, c, mean, S);
                    node_rough_normal(S, rough, normal);
                }
                out.rough[r] = rough;
#pragma unroll
                for (int k = 0; k < 3; ++k) { wm[lane * 3 + k] = mean[k]; wn[lane * 3 + k] = normal[k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) wc[lane * 6 + k] = (float)S[k];
            } else {
            const StageRow row = stage[src];            // (the 96-byte row in one piece: six 16-byte loads)
            out.sx[r] = row.sx; out.sy[r] = row.sy; out.sz[r] = row.sz;
            out.count[r] = row.count; out.first_idx[r] = row.first; out.flags[r] = row.flags;
            row_ncol[r] = row.idx_in_col == 0u ? row.ncol : 0u;
            float rough = 0.f, normal[3] = {0.f, 0.f, 0.f};
            if (row.flags & 1u) node_rough_normal(row.scatter, rough, normal);
            out.rough[r] = rough;
#pragma unroll
            for (int k = 0; k < 3; ++k) { wm[lane * 3 + k] = row.mean[k]; wn[lane * 3 + k] = normal[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) wc[lane * 6 + k] = (float)row.scatter[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t rows_here = min(64u, n - rb);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t e = (uint32_t)(k * 64 + lane);
            if (e < rows_here * 3u) { out.mean[3 * (size_t)rb + e] = wm[e]; out.normal[3 * (size_t)rb + e] = wn[e]; }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t e = (uint32_t)(k * 64 + lane);
            if (e < rows_here * 6u) out.cov[6 * (size_t)rb + e] = wc[e];
        }
        __builtin_amdgcn_wave_barrier();                // (the next iteration writes the same LDS)
    }
}

}  // namespace gndt
