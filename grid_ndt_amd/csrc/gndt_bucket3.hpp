// gndt_bucket3.hpp — k_bucket_direct: one workgroup per bucket.
//
//   accumulate (no barrier, no chunking: every wave streams its share of the bucket's records)
//       record -> key (divide-free exact index, gndt_math.hpp) -> number of its node (LDS index -> dense node arrays)
//              -> v = p - centre(node) in fp64 -> 9 x ds_add_f64 + count + first-seen straight into the node's sums
//   then, on the finished table (nodes are numbered in arrival order: the phases run over 0 .. n_nodes - 1, no compaction)
//       columns : column table; every node takes a number in its column, every column reserves its rows inside the bucket
//                 and gets an ARRAY of 16-byte node records (first-seen, z level, fp32 mean z)
//       rows    : slope label and index in column from a scan of the column's array; mean + fp64 scatter -> 96-B staging row,
//                 written WHOLE by one lane (a row written in two parts by two lanes cost twice the bytes at the memory side),
//                 a column's rows next to each other in first-seen order
//
// What bounds it (profiles/r03_bucket_ablation.txt, r03_c_sq.txt): ~350 vector instructions and ~15 LDS operations per point at
// four waves per SIMD — the vector port is ~56 % busy, the LDS pipeline about half, and the waves spend the rest waiting on
// each other (dependent LDS round trips queue behind the
// other waves' fp64 atomics, ~1 k cycles each under load).  More work in flight per thread (1 .. 4 records) changes nothing;
// what was tried against the probing cost (windows of slots, deferred probing, a second home slot, this index) and against
// the atomics (fixed point, integer atomics, bank binning) is listed there.  An earlier generation sorted every chunk of a
// bucket by node in LDS and summed runs in registers (~440 instructions per point, 242 us on the bench scene); a "thread pair
// owns a node" variant needed the same sort (238-256 us).  Hot buckets need no special casing; 64 or 512 identical consecutive
// points arrive as ONE weighted record (gndt_partition.hpp).  Tests run every strategy against the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_partition.hpp"

namespace gndt {

#ifndef GNDT_TALL_COLUMN
#define GNDT_TALL_COLUMN 24      // columns of more nodes than this find their z neighbours through the index (rows phase; 8 / 16 / 24 / 32 measured: S5 gains from 16 on, S2's 20-node columns lose below)
#endif
#ifndef GNDT_DIRECT_WAVES
#define GNDT_DIRECT_WAVES 4      // waves per SIMD (two 512-thread workgroups per CU): 128 VGPRs, no spills
#endif
#ifndef GNDT_DIRECT_THREADS
#define GNDT_DIRECT_THREADS 512  // threads of a workgroup with a 512-slot table
#endif
#ifndef GNDT_DIRECT_U
#define GNDT_DIRECT_U 2          // records per thread and iteration (512-slot table)
#endif

// Compact per-node statistics (the gndt_stats layout): what a shard of a multi-GPU build hands to the exchange.
// STATS = false: the bucket's nodes leave as staging rows (labels, moments) for the ordering + emit kernels.
// STATS = true : they leave as additive statistics (key, 9 sums, count, first index) and nothing else is done:
//                the shard's contribution to a global map (gndt_shard_stats_device).
struct StatsOut {
    uint64_t* key; double* sums; uint32_t* count; uint32_t* first;
};

// What the per-node phases need in LDS.  Round 6: it lives in the space of the nine sums, which every node's thread has taken into
// registers by then — 53 KB per workgroup instead of 70: THREE workgroups per CU (with one record per thread and iteration the
// accumulate loop fits the 80 registers that takes), and a workgroup in its phases leaves two, not one, accumulating beside it.
template <int H>
struct PhaseLds3 {
    uint4 colnodes[H];          // the nodes of every column as an ARRAY (column c at ccnt[c] >> 16): what a node's look at its column
                                //   needs of the others in one 16-byte load each, with independent addresses (loads pipeline; a linked
                                //   list chased one LDS round trip per node): first-seen index, z level, fp32 mean z (0 below min_points)
    float mz[H];                // (tall columns look their z neighbours up by node number)
    uint32_t chead[H];          // column table: a node of the column (its key is the column's key: no separate column keys), kNoNode = free
    uint32_t ccnt[H];           // nodes of the column in this column slot; after the prefix (first row inside the bucket) << 16 | nodes
};
template <int H>
struct BucketLds3 {           // 53 KB at H = 512
    // Node table in two parts.  `idx` is an open-addressing INDEX of 4 H words (at most a quarter full): the hash of a node's
    // key leads to a word holding a FINGERPRINT of the key and the number of the node.  Nodes are numbered in the order they
    // arrive, so keys and statistics sit in DENSE arrays: the per-node phases run over 0 .. n_nodes - 1 with no compaction
    // pass, and a table holds H nodes, not 0.78 H.
    alignas(16) uint32_t idx[4 * H];
    unsigned long long key[H];
    union {
        double sum[9][H];       // accumulate phase
        PhaseLds3<H> ph;        // per-node phases (the sums are in registers then)
    };
    uint32_t cnt[H];
    uint32_t first[H];
    uint32_t n_nodes, n_cols, n_slopes, stage_base, overflow, err_range, row_cursor;
    uint32_t clash;             // records whose fingerprint named another node (they went on with the key itself)
    uint32_t n_pairs;           // lanes whose two adjacent records fell into one node (what the host reads the cloud's locality from)
};
static_assert(sizeof(PhaseLds3<512>) <= sizeof(double) * 9 * 512, "the phase arrays fit the space of the sums");
// Workgroup barrier that orders LDS traffic ONLY (__syncthreads() also waits for the wave's global stores): the phases of
// the bucket kernels hand over LDS contents, their global stores are read by later kernels.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- finding a node: the index ----
// The index is H WINDOWS of four words.  A word is (fingerprint << 11) | nf: nf = 0 only in the EMPTY word (0); 1 .. 1024 = node
// number + 1; kNfPending = a lane has claimed the word and is about to publish the number of a NEW node; kNfFull = the table ran
// out of node numbers (the build is re-run).  The fingerprint is 21 bits of a second hash of the node key (never 0).  A node lives
// in the window its hash points at (any of its four words), or — that window full when it arrived — in the next one.
// ONE 16-byte read therefore answers "which node is this record's, and what is its number" for (nearly) every record of a wave:
// round 3's plain index took two DEPENDENT round trips per step (the word, then the 8-byte key it names; ~1 k cycles each under
// load, profiles/r03_bucket_ablation.txt) and a step per displaced slot, and with 128 records per wave some lane always needed
// three or four steps.  Here a window overflows with probability ~2e-4 (Poisson, 0.5 nodes per window) and a second step is rare.
// A fingerprint NAMES the node; the key confirms it: two keys that meet in a window share their fingerprint with probability
// 2^-21, so the key array is still read — once per record, requested after the search and looked at after the record's atomics
// have been issued.  A mismatch raises BucketLds3::clash: the bucket's table is cleared and filled again by accumulate_exact,
// which confirms every step with the key (~1 bucket in 10^4; tests narrow the fingerprint to force it in every bucket).
constexpr uint32_t kFpShift = 11u, kNfMask = 0x7FFu, kNfPending = 0x7FFu, kNfFull = 0x7FEu;
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
// the window of slot h (h & 3 = where a new node of this key looks for room first: spreads the claims of a window's keys)
template <typename Lds>
__device__ __forceinline__ uint4 lds_index_window(const Lds& L, uint32_t h) {
    asm volatile("" ::: "memory");                    // (other lanes write the index: every look is a new read)
    return *reinterpret_cast<const uint4*>(&L.idx[h & ~3u]);
}
// the word of window W that carries fingerprint fpw (0 if none does)
__device__ __forceinline__ uint32_t window_match(const uint4& W, uint32_t fpw) {
    uint32_t e = 0u;
    if ((W.w & ~kNfMask) == fpw) e = W.w;
    if ((W.z & ~kNfMask) == fpw) e = W.z;
    if ((W.y & ~kNfMask) == fpw) e = W.y;
    if ((W.x & ~kNfMask) == fpw) e = W.x;
    return e;
}
// first empty word of window W at or (cyclically) after position p0; 4 if the window is full
__device__ __forceinline__ uint32_t window_room(const uint4& W, uint32_t p0) {
    const uint32_t em = (W.x == 0u ? 1u : 0u) | (W.y == 0u ? 2u : 0u) | (W.z == 0u ? 4u : 0u) | (W.w == 0u ? 8u : 0u);
    if (em == 0u) return 4u;
    const uint32_t rot = ((em | (em << 4)) >> p0) & 0xFu;
    return (p0 + (uint32_t)__builtin_ctz(rot)) & 3u;
}

// The wave's lanes that claimed an empty word (`won`, the word at hw[j]) take node numbers — ONE addition per wave for all U
// records — and publish key and number.  id[j] / done[j] are set for the winners; a winner that finds the numbers used up marks
// the word kNfFull.
template <int H, int U, typename Lds>
__device__ __forceinline__ void lds_index_publish(Lds& L, const uint32_t (&hw)[U], const uint32_t (&fpw)[U], const unsigned long long (&key)[U],
                                                  const bool (&won)[U], uint32_t (&id)[U], bool (&use)[U], bool (&done)[U]) {
    unsigned long long m[U];
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) { m[j] = __ballot(won[j]); total += (uint32_t)__popcll(m[j]); }
    if (total == 0u) return;                          // (wave-uniform)
    uint32_t n0 = 0;
    if ((threadIdx.x & 63) == 0) n0 = atomicAdd(&L.n_nodes, total);
    n0 = (uint32_t)__shfl((int)n0, 0, 64);
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint32_t n = n0 + (uint32_t)__popcll(m[j] & below);
        n0 += (uint32_t)__popcll(m[j]);
        if (!won[j]) continue;
        if (n < (uint32_t)H) {
            L.key[n] = key[j];
            // The key must be in place before the number can be seen.  The LDS executes a wave's operations in order, so
            // only the COMPILER has to be kept from swapping the two stores: a signal fence, no s_waitcnt.
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(&L.idx[hw[j]], fpw[j] | (n + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            id[j] = n;
        } else {
            __hip_atomic_store(&L.idx[hw[j]], fpw[j] | kNfFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            L.overflow = 1; use[j] = false;
        }
        done[j] = true;
    }
}

// h[j]: where record j's search stands (updated); W[j]: the window it saw there (the caller's early look); id[j]: the node number
// once the search is over — NAMED by a fingerprint, to be confirmed by the caller with the key.  Records of a lane search
// together.  A record whose table is full gets use = false.  The usual case — every record of the wave finds its node in the
// window the hash points at — is settled by the early look without another LDS operation; only a wave that has to make room
// for a new node goes through the claim (compare-and-swap, numbers, publication).
template <int H, int U, typename Lds>
__device__ __forceinline__ void lds_index_find_or_insert(Lds& L, uint32_t (&h)[U], uint4 (&W)[U], uint32_t (&id)[U], bool (&use)[U],
                                                         const uint32_t (&fpw)[U], const unsigned long long (&key)[U]) {
    constexpr uint32_t kMask = 4u * (uint32_t)H - 1u;
    bool done[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { done[j] = !use[j]; id[j] = 0u; }
    for (int round = 0; round < 16 * H; ++round) {      // (a step settles a record, claims a word for it, moves it one window on, or waits for a publication)
        bool need = false, all_done = true;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (done[j]) continue;
            const uint32_t e = window_match(W[j], fpw[j]);
            const uint32_t nf = e & kNfMask;
            if (e == 0u) need = true;                     // not in this window: claim a word of it, or go on if it is full
            else if (nf < kNfFull) { id[j] = nf - 1u; done[j] = true; }
            else if (nf == kNfFull) { use[j] = false; done[j] = true; }
            // (kNfPending: the number is about to appear there — look again)
        }
        if (__any(need)) {
            bool won[U];
            uint32_t hw[U], old[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                won[j] = false; old[j] = 1u; hw[j] = 0u;
                if (!done[j] && window_match(W[j], fpw[j]) == 0u) {
                    const uint32_t p = window_room(W[j], h[j] & 3u);
                    if (p < 4u) { hw[j] = (h[j] & ~3u) | p; won[j] = true; old[j] = atomicCAS(&L.idx[hw[j]], 0u, fpw[j] | kNfPending); }
                    else h[j] = (h[j] + 4u) & kMask;          // the window is full: its neighbour
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) won[j] = won[j] && old[j] == 0u;      // (a loser looks at the window again: it may hold its own node now)
            lds_index_publish<H, U>(L, hw, fpw, key, won, id, use, done);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) all_done = all_done && done[j];
        if (__all(all_done)) return;                     // (the wave leaves together: the wave-wide operations above see all of its lanes)
#pragma unroll
        for (int j = 0; j < U; ++j) W[j] = lds_index_window(L, h[j]);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) if (!done[j]) { L.overflow = 1; use[j] = false; }
}

// The same search with every step confirmed by the 8-byte key, one record per lane: what accumulate_exact does for the bucket
// whose fingerprints clashed.  Called by whole waves (`active` per lane).  Returns the node's number, or kNoNode (table full).
template <int H, typename Lds>
__device__ __forceinline__ uint32_t lds_index_find_or_insert_exact(Lds& L, uint32_t h, const uint32_t fpw, const unsigned long long key, const bool active) {
    constexpr uint32_t kMask = 4u * (uint32_t)H - 1u;
    uint32_t node = kNoNode;
    bool done = !active;
    for (int round = 0; round < 16 * H; ++round) {
        if (__all(done)) break;
        if (done) continue;
        const uint4 W = lds_index_window(L, h);
        const uint32_t w4[4] = {W.x, W.y, W.z, W.w};
        bool wait = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = w4[i], nf = e & kNfMask;
            if (done || e == 0u || (e & ~kNfMask) != fpw) continue;
            if (nf == kNfPending) wait = true;            // (may be this key: look again once the number is there)
            else if (nf != kNfFull && L.key[nf - 1u] == key) { node = nf - 1u; done = true; }
        }
        if (done || wait) continue;
        const uint32_t p = window_room(W, h & 3u);
        if (p == 4u) { h = (h + 4u) & kMask; continue; }
        const uint32_t hw = (h & ~3u) | p;
        if (atomicCAS(&L.idx[hw], 0u, fpw | kNfPending) != 0u) continue;      // (somebody else took the word: look again)
        const uint32_t n = atomicAdd(&L.n_nodes, 1u);
        if (n < (uint32_t)H) {
            L.key[n] = key;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __hip_atomic_store(&L.idx[hw], fpw | (n + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            node = n;
        } else {
            __hip_atomic_store(&L.idx[hw], fpw | kNfFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            L.overflow = 1;
        }
        done = true;
    }
    if (!done) L.overflow = 1;
    return node;
}

// Look a key up in the finished index (no insertion): the node's number, or kNoNode.
template <int H, typename Lds>
__device__ __forceinline__ uint32_t lds_index_find(const Lds& L, uint32_t h, const uint32_t fpw, const unsigned long long key) {
    constexpr uint32_t kMask = 4u * (uint32_t)H - 1u;
    for (int probe = 0; probe < H; ++probe) {
        const uint4 W = *reinterpret_cast<const uint4*>(&L.idx[h & ~3u]);
        const uint32_t w4[4] = {W.x, W.y, W.z, W.w};
        bool room = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = w4[i], n = (e & kNfMask) - 1u;
            if (e == 0u) room = true;
            else if ((e & ~kNfMask) == fpw && n < (uint32_t)H && L.key[n] == key) return n;
        }
        if (room) return kNoNode;                         // (a window with room left never sent a node on to its neighbour)
        h = (h + 4u) & kMask;
    }
    return kNoNode;
}

// slot of a node inside its bucket's table.  The bucket is chosen by the TOP bits of the column hash (bucket_of), so its
// low bits are still uniform inside a bucket; the z level is spread over them with a full-rate 24-bit multiply.
__device__ __forceinline__ uint32_t node_slot3(uint32_t colh, int sz) {
    uint32_t g = colh ^ __umul24((uint32_t)sz & 0xFFFFFFu, 0x9E3779u);
    g ^= g >> 11;
    return g;
}

// Where a node's search starts in the index (low 11 bits of `slot`, masked by the caller) and its fingerprint word, from ONE short
// chain over the key (round 5; the bucket kernel is bound by vector-instruction issue, so every instruction per record counts):
// three full-rate 24-bit multiply-adds, a fold, and one more 24-bit multiply each for the slot and for the fingerprint — 13
// instructions where column hash + node_slot3 + node_fp_word took 27 and a quarter-rate 32-bit multiply.  The two outputs come from
// different halves of the folded word (a single 24-bit product carries 24 bits of entropy for 9 window + 21 fingerprint bits:
// measured 6-15 fingerprint clashes per build; with the second product 0-1, like the old chain).  Windows with more than four nodes,
// share of all windows, old chain | this one: bench scene 3.7e-3 | 3.2e-4, S3 6.3e-4 | 4.8e-4, S5 1.5e-3 | 6.4e-4.
__device__ __forceinline__ void node_index_hash(int sx, int sy, int sz, uint32_t fp_mask, uint32_t& slot, uint32_t& fpw) {
    const uint32_t a = (uint32_t)(sx + 65536), b = (uint32_t)(sy + 65536);      // (|sx|, |sy| <= 65535: 18 bits; the multiplies read 24)
    uint32_t t = __umul24(a, 0x9E3779u) + __umul24(b, 0x85EBCBu) + __umul24((uint32_t)sz, 0xC2B2AFu);
    t ^= t >> 15;
    uint32_t u = __umul24(t, 0x6D2B79u);
    const uint32_t f = __umul24(t >> 8, 0xA24BAFu);
    u ^= u >> 13;
    slot = u;
    fpw = max((f >> 11) & fp_mask, 1u) << kFpShift;                              // (fp_mask = 2^21 - 1 unless a test narrows it)
}

// One axis of a record's key and of its offset from the node's centre, from c = ceilf(|p - o| / len) however obtained: the signed
// index s (axis_from_ceil's) and v = p - (o + (2 s -+ 1) half_len) (axis_centre's, the same single fma).  Spelled for the
// instruction count: clamp by one median-of-three, 2 c - 1 and the sign in fp32 (exact: c <= 2^21), no integer detour.
__device__ __forceinline__ void axis_index_offset(float p, float o, float c, float limit, double half_len, double o64, bool& ok, int& s, double& v) {
    ok = ok && (c <= limit);                                         // (NaN: false — the record is dropped and counted)
    const float cc = __builtin_amdgcn_fmed3f(c, 1.0f, limit);       // 0 -> 1, beyond the key range -> the limit (ok is false then)
    const float k = fmaf(cc, 2.0f, -1.0f);
    const bool pos = p > o;                                          // (strict, as the reference: p == o and p = -0 go to the negative side)
    s = (int)(pos ? cc : -cc);
    v = (double)p - fma((double)(pos ? k : -k), half_len, o64);
}

template <int T, int H, typename Lds>
__device__ __forceinline__ void bucket_tables_init(Lds& L) {
    const int tid = threadIdx.x;
    for (int s = tid; s < H; s += T) {
        *reinterpret_cast<uint4*>(&L.idx[4 * s]) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 9; ++j) L.sum[j][s] = 0.0;
        L.cnt[s] = 0; L.first[s] = 0xFFFFFFFFu;
    }
    if (tid == 0) { L.n_nodes = 0; L.n_cols = 0; L.n_slopes = 0; L.stage_base = 0; L.overflow = 0; L.err_range = 0; L.row_cursor = 0; L.clash = 0; L.n_pairs = 0; }
}

// The accumulate phase once more for a bucket whose fingerprints clashed: one record per thread and step, every step of the
// search confirmed by the key.  Plain and slow on purpose — it runs for about one bucket in 10^4.
template <int T, int H, typename Lds>
__device__ __forceinline__ void accumulate_exact(Lds& L, const float4* __restrict__ recs, const uint32_t lo, const uint32_t hi,
                                                 const GridParams& P, const uint32_t fp_mask) {
    const double hx = 0.5 * (double)P.grid_len, hz = 0.5 * (double)P.z_len;
    const double ox = (double)P.ox, oy = (double)P.oy, oz = (double)P.oz;
    for (uint32_t base = lo; base < hi; base += (uint32_t)T) {
        if (__builtin_amdgcn_readfirstlane((int)L.overflow)) break;
        const uint32_t i = base + threadIdx.x;
        bool use = i < hi;
        const float4 rec = recs[min(i, hi - 1u)];
        const PointKey k = point_key_fast(rec.x, rec.y, rec.z, P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
        const unsigned long long pkey = pack_key(k.sx, k.sy, k.sz);
        uint32_t hs, hf;
        node_index_hash(k.sx, k.sy, k.sz, fp_mask, hs, hf);
        if (use && !k.ok) { atomicAdd(&L.err_range, 1u); use = false; }
        const uint32_t s = lds_index_find_or_insert_exact<H>(L, hs & (4u * (uint32_t)H - 1u), hf, pkey, use);
        if (!use || s == kNoNode) continue;
        const uint32_t iw = __float_as_uint(rec.w);
        const double v0 = (double)rec.x - fma((double)(2 * k.sx - (k.sx > 0 ? 1 : -1)), hx, ox);
        const double v1 = (double)rec.y - fma((double)(2 * k.sy - (k.sy > 0 ? 1 : -1)), hx, oy);
        const double v2 = (double)rec.z - fma((double)(2 * k.sz - (k.sz > 0 ? 1 : -1)), hz, oz);
        const double wf = (double)record_weight(iw);
        const double w0 = wf * v0, w1 = wf * v1, w2 = wf * v2;
        atomicAdd(&L.sum[0][s], w0); atomicAdd(&L.sum[1][s], w1); atomicAdd(&L.sum[2][s], w2);
        atomicAdd(&L.sum[3][s], w0 * v0); atomicAdd(&L.sum[4][s], w0 * v1); atomicAdd(&L.sum[5][s], w0 * v2);
        atomicAdd(&L.sum[6][s], w1 * v1); atomicAdd(&L.sum[7][s], w1 * v2); atomicAdd(&L.sum[8][s], w2 * v2);
        atomicAdd(&L.cnt[s], record_weight(iw));
        atomicMin(&L.first[s], record_index(iw));
    }
}

template <int T, int H, bool STATS, int UREC>
__device__ __forceinline__ void bucket_direct_one(BucketLds3<H>& L, const uint32_t bucket, const float4* __restrict__ recs,
                                                  const uint32_t lo, const uint32_t hi, const GridParams& P,
                                                  RawNode* __restrict__ stage, uint32_t stage_cap,
                                                  uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx,
                                                  const ColumnOrder& O, Counters* __restrict__ cnt,
                                                  PartCounters* __restrict__ pc, unsigned long long* __restrict__ dbg,
                                                  const StatsOut& so, const uint32_t fp_mask, const uint32_t interleave,
                                                  uint32_t* __restrict__ retry_list) {
    static_assert(H <= 65535, "node numbers are kept in 16 bits");
    constexpr int U = UREC;                            // records per thread and iteration
    const int tid = threadIdx.x;
    const int lane = tid & 63;
#define GNDT_STAMP3(k) do { if (dbg && tid == 0) dbg[(size_t)bucket * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    GNDT_STAMP3(0);
    bucket_tables_init<T, H>(L);
    __syncthreads();
    GNDT_STAMP3(1);

    // ---- accumulate: no barrier until every record of the bucket is in the table ----
    const double hx = 0.5 * (double)P.grid_len, hz = 0.5 * (double)P.z_len;
    const double ox = (double)P.ox, oy = (double)P.oy, oz = (double)P.oz;
    // A thread takes U = 2 ADJACENT records per iteration (32 bytes, a wave reads 2 KB in one piece).  Clouds with locality
    // (scan-ordered LiDAR, the zero padding) keep their order inside a bucket, so neighbouring records often belong to the
    // same node: the pair is then added as ONE contribution, and a wave whose 128 records all fall into one node adds them
    // with one set of atomics.  The records of the next iteration are loaded while this one is keyed and accumulated;
    // out-of-range lanes re-read the bucket's last record instead of branching around the load.
#ifdef GNDT_DIRECT_SUBSTAMPS      // diagnostic build: wave 0's clock at the stages of the loop, every stage drained before it is read
    unsigned long long st_acc[4] = {0ull, 0ull, 0ull, 0ull}, st0 = __builtin_amdgcn_s_memtime();
#define GNDT_SUB(k) do { if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st0; st0 = t_; } } while (0)
#else
#define GNDT_SUB(k) do { } while (0)
#endif
    // Which records a lane takes.  `interleave`: the lanes of ONE wave hold record pairs T / 64 pairs apart, the pairs in between go
    // to the other waves.  Clouds with locality keep their order inside a bucket, so records next to each other in the stream
    // belong to the same node more often than not (S3 / S5: half of all records follow one of their own node, a window of 128
    // records holds ~35 nodes); with consecutive pairs in consecutive lanes those meet inside ONE LDS atomic instruction, which
    // then serialises on the address: bucket kernel S3 32 M 681 -> 584 us, S5 355 -> 316 us (round 5).  A cloud without locality
    // (the bench scene) gains nothing and pays ~2 % for loads that cover 64 half-used lines each (the other waves take the other
    // halves), so the host turns it off once a build has counted (PartCounters::pairs) that adjacent records rarely share a node.
    // interleave == 2: every LANE owns a contiguous stretch of the bucket (the waves and the iterations walk through it), so the
    // records of one wave instruction are a 64th of the bucket apart — also runs of more than 16 records of one node (near-sensor
    // cells, the padding) stay in one lane and never meet inside an instruction.  All three mappings are
    //     pair(it, wave, lane) = lane * A + it * B + wave * C
    constexpr uint32_t W = (uint32_t)(T / 64);
    const uint32_t n_rec = hi - lo, iters = (n_rec + (uint32_t)(U * T) - 1u) / (uint32_t)(U * T);
    const uint32_t mapA = interleave == 0u ? 1u : (interleave == 1u ? W : iters * W), mapB = interleave == 2u ? W : (uint32_t)T,
                   mapC = interleave == 0u ? 64u : 1u;
    const uint32_t pair0 = (uint32_t)lane * mapA + (uint32_t)(tid >> 6) * mapC;      // this thread's pair in iteration 0
    uint32_t npairs = 0;                                    // (wave-uniform) lanes whose two records fell into one node
    float4 nxt[U];
    if (lo < hi) {
#pragma unroll
        for (int j = 0; j < U; ++j) nxt[j] = recs[min(lo + (uint32_t)U * pair0 + (uint32_t)j, hi - 1u)];
    }
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t mine = lo + (uint32_t)U * (pair0 + it * mapB);       // this thread's first record of the iteration
        // A table beyond its fill limit is given up at once: the build is re-run with more room anyway, and probing a nearly
        // full table costs hundreds of rounds per record (a cloud whose tables ALL overflow kept this kernel busy for 18-37 ms).
        if (__builtin_amdgcn_readfirstlane((int)L.n_nodes) > H) break;
        float4 rec[U];
        bool use[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { rec[j] = nxt[j]; use[j] = mine + (uint32_t)j < hi; }
        if (it + 1u < iters) {                     // uniform
#pragma unroll
            for (int j = 0; j < U; ++j) nxt[j] = recs[min(mine + (uint32_t)U * mapB + (uint32_t)j, hi - 1u)];
        }
        uint32_t slot[U], fpw[U];
        unsigned long long pkey[U];
        double v0[U], v1[U], v2[U];                   // the record's offset from its node's centre
#pragma unroll
        for (int j = 0; j < U; ++j) {
            // point_key_fast + pack_key + axis_centre, fused (same values bit for bit; ~40 vector instructions fewer per record)
            bool und = false;
            float cx = axis_ceil_try(rec[j].x, P.ox, P.inv_grid, und);
            float cy = axis_ceil_try(rec[j].y, P.oy, P.inv_grid, und);
            float cz = axis_ceil_try(rec[j].z, P.oz, P.inv_z, und);
            if (und) {                                 // (rare: within ~2 ulp of a cell border the reference's own divide decides)
                cx = ceilf(fabsf(rec[j].x - P.ox) / P.grid_len);
                cy = ceilf(fabsf(rec[j].y - P.oy) / P.grid_len);
                cz = ceilf(fabsf(rec[j].z - P.oz) / P.z_len);
            }
            bool ok = true;
            int sx, sy, sz;
            axis_index_offset(rec[j].x, P.ox, cx, (float)kMaxXY, hx, ox, ok, sx, v0[j]);
            axis_index_offset(rec[j].y, P.oy, cy, (float)kMaxXY, hx, oy, ok, sy, v1[j]);
            axis_index_offset(rec[j].z, P.oz, cz, (float)kMaxZ, hz, oz, ok, sz, v2[j]);
            pkey[j] = pack_key(sx, sy, sz);
            node_index_hash(sx, sy, sz, fp_mask, slot[j], fpw[j]);
            slot[j] &= 4u * (uint32_t)H - 1u;                                     // (where the search starts in the index)
            if (use[j] && !ok) { atomicAdd(&L.err_range, 1u); use[j] = false; }   // |nz| beyond the key range (x, y: the partition)
        }
        // an early look at the index: the window the hash points at is requested now and looked at after the arithmetic below — the
        // usual case (the node is there: a word carries its fingerprint and its number) costs no wait at all
        uint4 W0[U];
#ifndef GNDT_ABLATE_DIRECT
#pragma unroll
        for (int j = 0; j < U; ++j) W0[j] = lds_index_window(L, slot[j]);
#endif
        const bool pair = U >= 2 && use[0] && use[U - 1] && pkey[0] == pkey[U - 1];      // both records in one node: one contribution
        if (pair) use[U - 1] = false;
        npairs += (uint32_t)__popcll(__ballot(pair));
        // a wave whose records all sit in ONE node (dense cells, the zero padding): summed across the wave, one lane adds
        const bool one_node = __all(U >= 2 ? pair : use[0]) && __all(pkey[0] == __shfl(pkey[0], 0, 64));
        if (one_node) use[0] = lane == 0;
        double c[U][9];
        uint32_t cn[U], cf[U];
        // weighted records (64 or 512 identical points in one: the converters' zero padding) exist in few clouds and, there, in few
        // waves: a wave without one skips the decoding and the three multiplications by the weight (x 1.0 is exact: same sums)
        bool any_weight = false;
#pragma unroll
        for (int j = 0; j < U; ++j) any_weight = any_weight || (__float_as_uint(rec[j].w) & kWeight64Flag) != 0u;
        any_weight = __any(any_weight) != 0;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t iw = __float_as_uint(rec[j].w);
            double w0 = v0[j], w1 = v1[j], w2 = v2[j];
            cn[j] = 1u; cf[j] = iw;
            if (any_weight) {                                     // (wave-uniform)
                cn[j] = record_weight(iw);                        // exact: powers of two
                cf[j] = record_index(iw);
                const double wf = (double)cn[j];
                w0 = wf * v0[j]; w1 = wf * v1[j]; w2 = wf * v2[j];
            }
            c[j][0] = w0; c[j][1] = w1; c[j][2] = w2;
            c[j][3] = w0 * v0[j]; c[j][4] = w0 * v1[j]; c[j][5] = w0 * v2[j]; c[j][6] = w1 * v1[j]; c[j][7] = w1 * v2[j]; c[j][8] = w2 * v2[j];
        }
        if (pair) {
#pragma unroll
            for (int q = 0; q < 9; ++q) c[0][q] += c[U - 1][q];
            cn[0] += cn[U - 1]; cf[0] = min(cf[0], cf[U - 1]);
        }
        if (one_node) {
#pragma unroll
            for (int q = 0; q < 9; ++q) c[0][q] = wave_sum(c[0][q]);          // (lane 0 holds the total)
            for (int off = 32; off > 0; off >>= 1) { cn[0] += (uint32_t)__shfl_down((int)cn[0], off, 64); cf[0] = min(cf[0], (uint32_t)__shfl_down((int)cf[0], off, 64)); }
        }
        GNDT_SUB(0);
        uint32_t node[U];
        unsigned long long kv[U];
#ifdef GNDT_ABLATE_DIRECT      // TIMING ONLY (wrong map): the node's slot computed, not searched — the bound of a directly addressed table
#pragma unroll
        for (int j = 0; j < U; ++j) { node[j] = slot[j] >> 2; kv[j] = pkey[j]; }
#else
        lds_index_find_or_insert<H, U>(L, slot, W0, node, use, fpw, pkey);
        // the fingerprint NAMED the node; the key confirms it: requested now, looked at when the atomics are on their way
#pragma unroll
        for (int j = 0; j < U; ++j) kv[j] = L.key[use[j] ? node[j] : 0u];
#endif
        GNDT_SUB(it == 0u ? 1 : 2);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t s = node[j];
            const bool u = use[j];
            if (u) {
                atomicAdd(&L.sum[0][s], c[j][0]); atomicAdd(&L.sum[1][s], c[j][1]); atomicAdd(&L.sum[2][s], c[j][2]);
                atomicAdd(&L.sum[3][s], c[j][3]); atomicAdd(&L.sum[4][s], c[j][4]); atomicAdd(&L.sum[5][s], c[j][5]);
                atomicAdd(&L.sum[6][s], c[j][6]); atomicAdd(&L.sum[7][s], c[j][7]); atomicAdd(&L.sum[8][s], c[j][8]);
                atomicAdd(&L.cnt[s], cn[j]);
                atomicMin(&L.first[s], cf[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (use[j] && kv[j] != pkey[j]) L.clash = 1;       // the sums just went to ANOTHER node: this table is given up (below)
        GNDT_SUB(3);
    }
#ifdef GNDT_DIRECT_SUBSTAMPS
    if (dbg && tid == 0) { for (int k = 0; k < 4; ++k) dbg[(size_t)bucket * 16 + 8 + k] = st_acc[k]; }
#endif
    if (lane == 0 && npairs) atomicAdd(&L.n_pairs, npairs);
    __syncthreads();
    if (L.clash && !L.overflow) {            // uniform; ~1 bucket in 10^4: a fingerprint named the wrong node and the sums went there
        __syncthreads();                     // (everybody has read the flags)
        bucket_tables_init<T, H>(L);
        __syncthreads();
        accumulate_exact<T, H>(L, recs, lo, hi, P, fp_mask);
        if (tid == 0) atomicAdd(&pc->fp_clashes, 1u);
        __syncthreads();
    }
    GNDT_STAMP3(2);
#ifdef GNDT_ABLATE_NO_PHASES   // TIMING ONLY (no map): what the kernel costs without its per-node phases
    return;
#endif
    const uint32_t M = L.n_nodes;
    if (L.overflow || M > (uint32_t)H) {         // uniform
        // More nodes than the table has numbers.  Nothing of this bucket has left the workgroup yet, so it can simply be done again:
        // with a retry list (round 5) the bucket is queued for the second pass — the same kernel with 1024-slot tables over the
        // listed buckets only — and the build goes on; without one (that second pass itself, or a build launched without it) the
        // flag is raised and the host re-runs the whole build with more room, as before.
        if (tid == 0) {
            if (retry_list) retry_list[atomicAdd(&pc->lds_retry, 1u)] = bucket;
            else atomicAdd(&pc->lds_overflow, 1u);
        }
        return;
    }
    // Reserve the staging rows now.  The memory-side atomic's round trip (~3 us) hides behind the column phases: its answer stays in
    // the register of the thread that asked and is handed to the workgroup only at the barrier in front of the rows phase, where it is
    // first needed.  (Rounds 1-3 said the same in this comment and then waited for the answer at the very next barrier — ~5 k cycles per
    // bucket that the stamps booked under "columns"; on a small frame, one workgroup per CU, nothing else ran meanwhile.)
    uint32_t stage_base_reg = 0;
    if (tid == T - 1) stage_base_reg = atomicAdd(&cnt->num_nodes, M);
    if (tid == 0 && L.err_range) atomicAdd(&cnt->err_key_range, L.err_range);
    if (tid == 64 && L.n_pairs) atomicAdd(&pc->pairs, L.n_pairs);

    if constexpr (STATS) {
        if (tid == T - 1) L.stage_base = stage_base_reg;
        __syncthreads();
        const uint32_t sbase = L.stage_base;
        if (sbase + M > stage_cap) {                   // uniform
            if (tid == 0) atomicAdd(&pc->stage_overflow, M);
            return;
        }
        for (uint32_t i = tid; i < M; i += T) {
            const uint32_t s = i;
            const uint32_t dst = sbase + i;
            so.key[dst] = L.key[s];
#pragma unroll
            for (int j = 0; j < 9; ++j) so.sums[9 * (size_t)dst + j] = L.sum[j][s];
            so.count[dst] = L.cnt[s];
            so.first[dst] = L.first[s];
        }
        return;
    }

    // ---- per-node phases: ONE node per thread (a table holds at most H = T nodes), so what a node learns in one phase — its
    //      key, count, first-seen index, mean z, column slot, number in its column — stays in REGISTERS across the barriers
    //      (round 3 kept it in LDS arrays: three more dependent round trips, each behind the other workgroups' fp64 atomics) ----
    // Round 6: so do its nine SUMS.  Every thread takes its node's statistics out of the table first; from the barrier behind that
    // on, the space of the sums holds the phases' own arrays (BucketLds3::ph), and the node leaves as a RawNode record — statistics
    // as they are; mean, scatter and eigen-solve are k_emit_rows' — written whole by its thread, a column's rows next to each other.
    static_assert(H <= T, "one node per thread");
    const bool live = (uint32_t)tid < M;
    const uint32_t s = (uint32_t)tid;            // (node s in thread s.  Transposed over the live waves — the levels of a column arrive together
                                                 //  and would otherwise meet in one LDS atomic on the column's counter — measured in round 5: no change,
                                                 //  columns 4.9 k / 7.0 k cycles per bucket on S2 / S3 either way: these phases are round trips, not conflicts)
    uint64_t key = 0;
    int sx = 0, sy = 0, sz = 0;
    uint32_t my_n = 0, my_first = 0xFFFFFFFFu, col = 0, kc = 0;
    float cz = 0.f;
    double sums[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (live) {
        key = L.key[s];
        my_n = L.cnt[s];
        my_first = L.first[s];
#pragma unroll
        for (int j = 0; j < 9; ++j) sums[j] = L.sum[j][s];
    }
    lds_barrier();                               // (every sum is in a register: their space is the phases' from here on)
    for (int c = tid; c < H; c += T) { L.ph.chead[c] = kNoNode; L.ph.ccnt[c] = 0u; }
    if (tid == 0) { L.n_cols = 0; L.n_slopes = 0; L.row_cursor = 0; }
    if (live) {
        unpack_key(key, sx, sy, sz);
        cz = (my_n >= (uint32_t)P.min_points) ? node_mean_z(my_n, sums[2], axis_centre(sz, P.oz, P.z_len)) : 0.f;
        L.ph.mz[s] = cz;                                      // (tall columns look their z neighbours up by node number)
    }
    lds_barrier();
    // ---- columns: every node finds the slot of its column and takes a number in it ----
    if (live) {
        const uint64_t ck = column_key(key);
        col = ((column_hash(sx, sy) * 0x85EBCA77u) >> 12) & (uint32_t)(H - 1);
        for (int probe = 0; probe < H; ++probe) {            // (terminates: the table has more slots than nodes)
            uint32_t head = L.ph.chead[col];
            if (head == kNoNode) {
                head = atomicCAS(&L.ph.chead[col], kNoNode, s);
                if (head == kNoNode) { atomicAdd(&L.n_cols, 1u); break; }                           // first node of a new column
            }
            if (column_key(L.key[head]) == ck) break;
            col = (col + 1) & (uint32_t)(H - 1);
        }
        kc = atomicAdd(&L.ph.ccnt[col], 1u);
    }
    lds_barrier();
    // The rows of a column are staged NEXT TO EACH OTHER, in first-seen order: every column reserves its rows inside the bucket.
    // The ordering pass then works per column (one lookup of the column's place instead of one per node) and the emit pass
    // gathers runs of rows.  The same reservation places the columns' node arrays.
    for (int c = tid; c < H; c += T) {              // (one LDS atomic per column: the columns' order inside the bucket is free)
        const uint32_t v = L.ph.ccnt[c];
        if (v) L.ph.ccnt[c] = (atomicAdd(&L.row_cursor, v) << 16) | v;
    }
    lds_barrier();
    uint32_t cinfo = 0;
    if (live) {
        cinfo = L.ph.ccnt[col];
        L.ph.colnodes[(cinfo >> 16) + kc] = make_uint4(my_first, (uint32_t)sz, __float_as_uint(cz), s);
    }
    if (tid == T - 1) L.stage_base = stage_base_reg;      // (the reservation's answer: waited for here, by one thread)
    lds_barrier();
    GNDT_STAMP3(3);
    const uint32_t sbase = L.stage_base;
    if (sbase + M > stage_cap) {                   // uniform: the staging rows ran out, the build is re-run with more
        if (tid == 0) atomicAdd(&pc->stage_overflow, M);
        return;
    }

    // ---- rows: slope label (OcNode::isSlope, map2D.h:66-108), index in column, column size and first-seen index by
    //      walking the column's short array; the node's record ----
    uint32_t my_slopes = 0;
    if (live) {
        const int za = level_above(sz), zb = level_below(sz);
        uint32_t icol = 0, cf = 0xFFFFFFFFu;
        bool up = false, down = false;
        const uint32_t cbase = cinfo >> 16;
        const uint32_t ncol = cinfo & 0xFFFFu;
        uint32_t fl = (my_n >= (uint32_t)P.min_points) ? 1u : 0u;
        // Tall columns (walls: dozens of levels) look their two z neighbours up in the node index — two short probes — and walk
        // the column only for what needs every node of it (index in column, the column's first-seen index): three instructions
        // per node instead of ten.  Short columns find the neighbours during the walk, as before.
        const bool tall = ncol > (uint32_t)GNDT_TALL_COLUMN;
        if (tall) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int tz = side == 0 ? za : zb;
                uint32_t hs, hf;
                node_index_hash(sx, sy, tz, fp_mask, hs, hf);
                const uint32_t t = lds_index_find<H>(L, hs & (4u * (uint32_t)H - 1u), hf, pack_key(sx, sy, tz));
                if (t != kNoNode) {
                    const float oz2 = (L.first[t] < my_first) ? L.ph.mz[t] : 0.f;            // "visited": seen earlier AND has statistics
                    const bool far = fabsf(oz2 - cz) > P.slope_interval;
                    if (side == 0) up = far; else down = far;
                }
            }
            for (uint32_t k0 = 0; k0 < ncol; k0 += 4) {
                uint32_t tf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tf[j] = L.ph.colnodes[cbase + min(k0 + (uint32_t)j, ncol - 1u)].x;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + (uint32_t)j < ncol) { cf = min(cf, tf[j]); icol += (tf[j] < my_first) ? 1u : 0u; }
            }
        } else
        for (uint32_t k0 = 0; k0 < ncol; k0 += 4) {         // four independent loads in flight, then their four nodes
            uint4 rr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = L.ph.colnodes[cbase + min(k0 + (uint32_t)j, ncol - 1u)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 r = rr[j];                      // {first-seen, z level, mean z or 0, node slot}
                const uint32_t tf = r.x;
                if (k0 + (uint32_t)j < ncol && tf != my_first) {     // (first-seen indices of distinct nodes are distinct points)
                    cf = min(cf, tf);
                    icol += (tf < my_first) ? 1u : 0u;
                    const int tz = (int)r.y;
                    if (tz == za || tz == zb) {
                        const float oz2 = (tf < my_first) ? __uint_as_float(r.z) : 0.f;     // "visited": seen earlier AND has statistics
                        const bool far = fabsf(oz2 - cz) > P.slope_interval;
                        if (tz == za) up = up || far; else down = down || far;
                    }
                }
            }
        }
        cf = min(cf, my_first);
        if (fl) {
            bool slope = true;
            if (P.demand == 0) slope = !up; else down = false;
            if (slope) { fl |= 2u; if (down) fl |= 4u; ++my_slopes; }
        }
        RawNode row;
        row.key = key; row.count = my_n; row.first = my_first;
#pragma unroll
        for (int j = 0; j < 9; ++j) row.sum[j] = sums[j];
        row.info = fl | (icol << 3);
        row.ncol = ncol;
        const uint32_t dst = sbase + cbase + icol;
        stage[dst] = row;
        ord_cf[dst] = cf;
        ord_idx[dst] = icol ? icol : (kOrdHeadFlag | ncol);       // (a column's first row carries the column's size)
        if (icol == 0) note_column(O, cf, ncol);
    }
    // counters: aggregated in LDS, ONE memory-side atomic per bucket and counter
    if (my_slopes) atomicAdd(&L.n_slopes, my_slopes);
    lds_barrier();
    // (columns and slopes of the bucket in ONE atomic, on the partition counters' line — not on the line every bucket's row reservation waits on)
    if (tid == 0 && (L.n_cols | L.n_slopes)) atomicAdd(&pc->cols_slopes, ((unsigned long long)L.n_slopes << 32) | (unsigned long long)L.n_cols);
    GNDT_STAMP3(4);
#undef GNDT_STAMP3
}

// Bucket b, b + gridDim.x, ...  With 512-slot tables two 512-thread workgroups share a CU (61 KB of LDS each, four waves per
// SIMD); the 1024-slot variant (retries, node-heavy clouds) runs one 1024-thread workgroup per CU.
// UREC / WAVES (512-slot tables): two records per thread and iteration at four waves per SIMD (128 registers, two workgroups per CU:
// adjacent records of one node are added as ONE contribution — what clouds with locality live on), or one record at six waves per
// SIMD (80 registers, THREE workgroups per CU): the host takes the second for clouds whose neighbouring records rarely share a node.
template <int T, int H, bool STATS = false, int UREC = 2, int WAVES = 4>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_bucket_direct(const float4* __restrict__ recs, BucketRanges ranges, uint32_t num_buckets, GridParams P,
                                                     RawNode* __restrict__ stage, uint32_t stage_cap,
                                                     uint32_t* __restrict__ ord_cf, uint32_t* __restrict__ ord_idx, ColumnOrder O,
                                                     Counters* __restrict__ cnt, PartCounters* __restrict__ pc,
                                                     unsigned long long* __restrict__ dbg, StatsOut so, uint32_t fp_mask, uint32_t interleave,
                                                     uint32_t* __restrict__ retry_list, const uint32_t* __restrict__ todo_list) {
    // todo_list: the second pass — the buckets a first pass queued (pc->lds_retry of them), instead of all num_buckets
    __shared__ BucketLds3<H> L;
    const uint32_t count = todo_list ? min(pc->lds_retry, num_buckets) : num_buckets;
    if (!todo_list && blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t e = pc->l1_err; if (e) atomicAdd(&cnt->err_key_range, e); }   // (FoldClear, gndt_partition.hpp)
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t bucket = todo_list ? todo_list[i] : i;
        uint32_t lo, hi;
        bucket_range(ranges, bucket, lo, hi);
        bucket_direct_one<T, H, STATS, UREC>(L, bucket, recs, lo, hi, P, stage, stage_cap, ord_cf, ord_idx, O, cnt, pc, dbg, so, fp_mask, interleave, retry_list);
        lds_barrier();          // the LDS tables are re-initialised by the next bucket
    }
}

}  // namespace gndt
