// gndt_tile.hpp — strategy TILE: ONE pass over a cloud that keeps its scan order (BASELINE north_star: "coalesced HBM reads
// of XYZ ... LDS-staged privatised reductions into the 3x3 upper-triangle accumulators").
//
// Every workgroup owns a contiguous range of the cloud and a PRIVATE node table in LDS (the table of k_bucket_direct):
// points are keyed and added to it with LDS atomics, neighbouring points of one node first merged in registers.  When the
// table holds kTileFlush nodes (or the range ends) the workgroup flushes ONE partial per distinct node — key, 9 fp64 sums,
// count, first-seen index — into the persistent HBM table of strategy ATOMIC (find-or-insert + 11 memory-side atomics, the
// merge of k_stats_merge) and starts over.  Finalisation is the table path's (gndt_table.hpp), so incremental updates,
// statistics export and the multi-GPU merge work unchanged.
//
// What it costs: 12 B per point streamed once, and 11 memory-side atomics per (node, flush) instead of per point.  A
// memory-side atomic request is ~20 G/s chip-wide (MI355X_MICROARCH, Global float atomics), the partition pipeline
// ~18 G points/s, so this path wins when a flush carries well over a dozen points per node — dense scans, small voxels,
// the reference's zero padding — and loses on shuffled clouds, where every point is its own partial.  gndt_build* with
// strategy AUTO measures that ratio on a sample of the cloud (k_tile_sample) before choosing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_bucket3.hpp"

namespace gndt {

constexpr int kTileT = 512, kTileH = 512;
constexpr int kTileFlush = 224;             // nodes in the LDS table that trigger a flush (checked every kTileCheck points)
constexpr int kTileCheck = 2048;            // points a workgroup adds between two looks at the fill (no barrier inside)

struct TileLdsTable {
    unsigned long long key[kTileH];
    double sum[9][kTileH];
    uint32_t cnt[kTileH];
    uint32_t first[kTileH];
    uint16_t list[kTileH];
    uint32_t n_nodes, n_list, err_range, err_full, bound, pad[3];
};

// One partial (a node's statistics from one tile) into the HBM table.  Called by whole waves (`live` may differ per lane):
// the node-list append is wave-aggregated.
__device__ __forceinline__ void table_merge_partial(bool live, uint64_t key, const double (&q)[9], uint32_t count, uint32_t first,
                                                    uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                                    uint32_t* __restrict__ node_slot, uint32_t* __restrict__ index_of_slot,
                                                    uint32_t* __restrict__ touch_epoch, uint32_t* __restrict__ touched, int mark,
                                                    uint32_t epoch, Counters* __restrict__ cnt) {
    bool inserted = false;
    uint32_t slot = cap_mask + 1;
    if (live) {
        slot = find_or_insert(keys, cap_mask, key, inserted);
        if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); inserted = false; }
    }
    append_new_nodes(inserted, slot, node_slot, index_of_slot, cnt);
    const bool ok = live && slot <= cap_mask;
    if (mark) {                                      // (wave-uniform) incremental update: list the node as touched once per frame
        const bool fresh = ok && atomicExch(&touch_epoch[slot], epoch) != epoch;
        const unsigned long long m = __ballot(fresh);
        if (m) {
            const int lane = threadIdx.x & 63, leader = (int)__builtin_ctzll(m);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&cnt->n_touched, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader, 64);
            if (fresh) touched[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
        }
    }
    if (ok) {
        NodeAcc* a = acc + slot;
#pragma unroll
        for (int k = 0; k < 9; ++k) unsafeAtomicAdd(&a->s[k], q[k]);
        atomicAdd(&a->count, count);
        atomicMin(&a->first, first);
    }
}

// Flush the workgroup's LDS table into the HBM table and leave it empty.  All threads call it (barriers inside).
template <typename Lds>
__device__ __forceinline__ void tile_flush(Lds& L, uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                           uint32_t* __restrict__ node_slot, uint32_t* __restrict__ index_of_slot,
                                           uint32_t* __restrict__ touch_epoch, uint32_t* __restrict__ touched, int mark, uint32_t epoch,
                                           Counters* __restrict__ cnt) {
    const int tid = threadIdx.x, lane = tid & 63;
    // compact the occupied slots
    {
        const bool occ = L.key[tid] != kEmptyKey;
        const unsigned long long m = __ballot(occ);
        uint32_t wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&L.n_list, (uint32_t)__popcll(m));
        wbase = (uint32_t)__shfl((int)wbase, 0, 64);
        if (occ) L.list[wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    }
    __syncthreads();
    const uint32_t M = L.n_list;
    const uint32_t M_round = (M + 63u) & ~63u;         // whole waves: the list append inside is wave-aggregated
    for (uint32_t i = tid; i < M_round; i += kTileT) {
        const bool live = i < M;
        const uint32_t s = live ? (uint32_t)L.list[i] : 0u;
        double q[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) q[j] = L.sum[j][s];
        table_merge_partial(live, L.key[s], q, L.cnt[s], L.first[s], keys, acc, cap_mask, node_slot, index_of_slot, touch_epoch, touched,
                            mark, epoch, cnt);
        if (live) {                                    // the slot is free again
            L.key[s] = kEmptyKey;
#pragma unroll
            for (int j = 0; j < 9; ++j) L.sum[j][s] = 0.0;
            L.cnt[s] = 0; L.first[s] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    if (tid == 0) { L.n_nodes = 0; L.n_list = 0; }
    __syncthreads();
}

template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kTileT) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_tile_accumulate(const float* __restrict__ xyz, uint64_t n, uint32_t first_base, int base_from_device, GridParams P,
                  uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask, uint32_t* __restrict__ node_slot,
                  uint32_t* __restrict__ index_of_slot, uint32_t* __restrict__ touch_epoch, uint32_t* __restrict__ touched, int mark,
                  Counters* __restrict__ cnt) {
    constexpr int T = kTileT, H = kTileH;
    __shared__ TileLdsTable L;
    const int tid = threadIdx.x, lane = tid & 63;
    L.key[tid] = kEmptyKey;
#pragma unroll
    for (int j = 0; j < 9; ++j) L.sum[j][tid] = 0.0;
    L.cnt[tid] = 0; L.first[tid] = 0xFFFFFFFFu;
    if (tid == 0) { L.n_nodes = 0; L.n_list = 0; L.err_range = 0; L.err_full = 0; L.bound = 0; }
    __syncthreads();
    const uint32_t epoch = cnt->epoch;
    const uint32_t base_idx = base_from_device ? cnt->stream_pos : first_base;
    // Which points a lane takes: the lanes of one wave hold point pairs T / 64 pairs apart, the other waves the pairs in between
    // (gndt_bucket3.hpp, round 5).  This kernel only runs on clouds whose neighbours in the stream share their node — a depth
    // frame's rows —, where consecutive lanes sent the same address into one LDS atomic instruction: depth frame 0.059 -> 0.054 ms
    // per build, 0.048 -> 0.042 replayed.
#ifndef GNDT_TILE_INTERLEAVE
#define GNDT_TILE_INTERLEAVE 1
#endif
    const uint32_t ptid = GNDT_TILE_INTERLEAVE ? (uint32_t)lane * (uint32_t)(T / 64) + (uint32_t)(tid >> 6) : (uint32_t)tid;
    // this workgroup's contiguous range, a multiple of kTileCheck points
    uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + kTileCheck - 1) / kTileCheck * kTileCheck;
    const uint64_t lo = min(n, (uint64_t)blockIdx.x * per), hi = min(n, lo + per);
    const double hx = 0.5 * (double)P.grid_len, hz = 0.5 * (double)P.z_len;
    const double ox = (double)P.ox, oy = (double)P.oy, oz = (double)P.oz;
    for (uint64_t t0 = lo; t0 < hi; t0 += kTileCheck) {
        const uint32_t have = (uint32_t)min((uint64_t)kTileCheck, hi - t0);
        const float* __restrict__ src = xyz + t0 * STRIDE_FLOATS;           // uniform: 64-bit arithmetic stays scalar
        for (uint32_t b = 0; b < have; b += 2 * T) {                        // a thread takes two ADJACENT points per step
            float px[2], py[2], pz[2];
            bool use[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t i = b + 2u * ptid + j;
                use[j] = i < have;
                const float* p = src + min(i, have - 1u) * (uint32_t)STRIDE_FLOATS;
                px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
            }
            PointKey k[2];
            unsigned long long pkey[2];
            uint32_t slot[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                k[j] = point_key_fast(px[j], py[j], pz[j], P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
                pkey[j] = pack_key(k[j].sx, k[j].sy, k[j].sz);
                slot[j] = node_slot3(column_hash(k[j].sx, k[j].sy), k[j].sz) & (uint32_t)(H - 1);
                if (use[j] && !k[j].ok) { atomicAdd(&L.err_range, 1u); use[j] = false; }
            }
            const bool pair = use[0] && use[1] && pkey[0] == pkey[1];
            if (pair) use[1] = false;
            double c[2][9];
            uint32_t cn[2], cf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double v0 = (double)px[j] - fma((double)(2 * k[j].sx - (k[j].sx > 0 ? 1 : -1)), hx, ox);
                const double v1 = (double)py[j] - fma((double)(2 * k[j].sy - (k[j].sy > 0 ? 1 : -1)), hx, oy);
                const double v2 = (double)pz[j] - fma((double)(2 * k[j].sz - (k[j].sz > 0 ? 1 : -1)), hz, oz);
                c[j][0] = v0; c[j][1] = v1; c[j][2] = v2;
                c[j][3] = v0 * v0; c[j][4] = v0 * v1; c[j][5] = v0 * v2; c[j][6] = v1 * v1; c[j][7] = v1 * v2; c[j][8] = v2 * v2;
                cn[j] = 1u;
                cf[j] = base_idx + (uint32_t)(t0 + b + 2u * ptid + j);
            }
            if (pair) {
#pragma unroll
                for (int q = 0; q < 9; ++q) c[0][q] += c[1][q];
                cn[0] = 2u;
            }
            // a wave whose 128 points all sit in ONE node (dense cells, the zero padding): summed across the wave
            if (__all(pair) && __all(pkey[0] == __shfl(pkey[0], 0, 64))) {
#pragma unroll
                for (int q = 0; q < 9; ++q) c[0][q] = wave_sum(c[0][q]);
                for (int off = 32; off > 0; off >>= 1) { cn[0] += (uint32_t)__shfl_down((int)cn[0], off, 64); cf[0] = min(cf[0], (uint32_t)__shfl_down((int)cf[0], off, 64)); }
                use[0] = lane == 0;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!use[j]) continue;
                uint32_t s = slot[j];
                if (L.key[s] != pkey[j]) s = lds_find_or_insert<H>(L.key, slot[j], pkey[j], &L.n_nodes);
                if (s < (uint32_t)H) {
                    atomicAdd(&L.sum[0][s], c[j][0]); atomicAdd(&L.sum[1][s], c[j][1]); atomicAdd(&L.sum[2][s], c[j][2]);
                    atomicAdd(&L.sum[3][s], c[j][3]); atomicAdd(&L.sum[4][s], c[j][4]); atomicAdd(&L.sum[5][s], c[j][5]);
                    atomicAdd(&L.sum[6][s], c[j][6]); atomicAdd(&L.sum[7][s], c[j][7]); atomicAdd(&L.sum[8][s], c[j][8]);
                    atomicAdd(&L.cnt[s], cn[j]);
                    atomicMin(&L.first[s], cf[j]);
                } else {
                    // the private table is full (a tile without locality): this contribution goes to the HBM table directly.
                    // Lane by lane (no wave-aggregated append here: the lanes that get here differ from step to step).
                    bool inserted;
                    const uint32_t g = find_or_insert(keys, cap_mask, pkey[j], inserted);
                    if (g > cap_mask) { atomicAdd(&L.err_full, 1u); continue; }
                    if (inserted) { const uint32_t idx = atomicAdd(&cnt->num_nodes, 1u); node_slot[idx] = g; index_of_slot[g] = idx; }
                    if (mark && atomicExch(&touch_epoch[g], epoch) != epoch) touched[atomicAdd(&cnt->n_touched, 1u)] = g;
                    NodeAcc* a = acc + g;
#pragma unroll
                    for (int q = 0; q < 9; ++q) unsafeAtomicAdd(&a->s[q], c[j][q]);
                    atomicAdd(&a->count, cn[j]);
                    atomicMin(&a->first, cf[j]);
                }
            }
        }
        __syncthreads();
        if (L.n_nodes >= (uint32_t)kTileFlush || t0 + kTileCheck >= hi)       // uniform
            tile_flush(L, keys, acc, cap_mask, node_slot, index_of_slot, touch_epoch, touched, mark, epoch, cnt);
    }
    if (tid == 0) {
        if (L.err_range) atomicAdd(&cnt->err_key_range, L.err_range);
        if (L.err_full) atomicAdd(&cnt->err_table_full, L.err_full);
    }
}

// Locality probe for strategy AUTO: `tiles` tiles of kTileCheck consecutive points spread evenly over the cloud; for each, the
// number of DISTINCT nodes its points fall into (LDS key table only).  out[0] += points looked at, out[1] += distinct nodes
// (= partials a flush would send).  points / partials is what strategy TILE gains over one set of atomics per point.
// out: three device words {points, nodes, ticket}, zero between launches; host_out: two pinned words the last workgroup fills.
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kTileT) k_tile_sample(const float* __restrict__ xyz, uint64_t n, GridParams P, uint32_t tiles,
                                                        unsigned long long* __restrict__ out, unsigned long long* __restrict__ host_out) {
    constexpr int T = kTileT, H = 4096;              // a table a whole tile fits in whatever its locality
    __shared__ unsigned long long key[H];
    __shared__ uint32_t n_new;
    const int tid = threadIdx.x;
    for (int s = tid; s < H; s += T) key[s] = kEmptyKey;
    if (tid == 0) n_new = 0;
    __syncthreads();
    const uint64_t ntiles = (n + kTileCheck - 1) / kTileCheck;
    const uint64_t tile = (uint64_t)blockIdx.x * ntiles / tiles;
    const uint64_t t0 = tile * kTileCheck;
    const uint32_t have = t0 < n ? (uint32_t)min((uint64_t)kTileCheck, n - t0) : 0u;
    for (uint32_t i = tid; i < have; i += T) {
        const float* p = xyz + (t0 + i) * STRIDE_FLOATS;
        const PointKey k = point_key_fast(p[0], p[1], p[2], P.ox, P.oy, P.oz, P.grid_len, P.z_len, P.inv_grid, P.inv_z);
        if (!k.ok) continue;
        lds_find_or_insert<H>(key, node_slot3(column_hash(k.sx, k.sy), k.sz), pack_key(k.sx, k.sy, k.sz), &n_new);
    }
    __syncthreads();
    // The totals go home with the LAST workgroup (ticket out[2]): it stores them into the host's pinned words and zeroes the three
    // device words for the next sample — no memset in front of the kernel, no copy behind it (two stream operations of ~5-8 us each
    // on the path of a handle's first build, which waits for this answer: round 6).
    if (tid < 64) {                                    // (the first wave; lanes 0 and 1 carry one total each: their round trips overlap)
        const unsigned long long mine = tid == 0 ? (unsigned long long)have : (unsigned long long)n_new;
        if (tid < 2) atomicAdd(&out[tid], mine);
        __threadfence();                               // (both totals are in before the ticket is taken)
        int last = 0;
        if (tid == 0) last = atomicAdd(&out[2], 1ull) == (unsigned long long)gridDim.x - 1ull ? 1 : 0;
        last = __shfl(last, 0, 64);
        if (last) {
            if (tid < 2) host_out[tid] = atomicAdd(&out[tid], 0ull);
            if (tid < 3) out[tid] = 0ull;
        }
    }
}

}  // namespace gndt
