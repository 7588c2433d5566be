// gndt_exchange.hpp — the statistics exchange that builds ONE global map from a cloud sharded over the GPUs of a node
// (BASELINE.json configs[2]), with RCCL called from C++ behind the C ABI: a C++ / ROS host needs no Python
// and no torch.distributed to shard a cloud.  RCCL is resolved at run time (dlopen): single-GPU users of libgndt do not
// need it installed.
//
// Per-node statistics in cell-local coordinates are additive over any partition of the points and the first-seen index
// combines with min, so one round builds the global map:
//   all-gather   every rank's occupied keys (8 B each)  -> sort + unique on the device = the canonical node order
//   all-reduce   SUM over a PACKED [C x 10] fp64 buffer (9 sums + the count, exact in fp64) scattered into that order
//   all-reduce   MIN over [C] u32 first-seen indices
//   every rank   gndt_finalize_stats_device on the reduced statistics (already sorted by key: a column's nodes adjacent)
// Only occupied nodes travel (84 B each), never points.  xGMI is point to point (7 links per GPU): the ring all-reduce
// moves 2 (N-1)/N x 84 B x C per GPU, the dominant cost at scale.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rccl/rccl.h>          // types and prototypes only: the functions come from dlopen

#include "gndt_partition.hpp"

namespace gndt {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;      // (optional: an all-reduce of the whole array does the same job)
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllGather && AllReduce && GetErrorString; }
    bool p2p() const { return Send && Recv && GroupStart && GroupEnd; }
};

// the RCCL the process already has (PyTorch bundles its own librccl.so) or the system one
inline const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            a.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (a.lib) break;
        }
        if (!a.lib)
            for (const char* nm : names) {
                a.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
                if (a.lib) break;
            }
        if (a.lib) {
            a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
            a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
            a.AllGather = (decltype(a.AllGather))dlsym(a.lib, "ncclAllGather");
            a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
            a.ReduceScatter = (decltype(a.ReduceScatter))dlsym(a.lib, "ncclReduceScatter");
            a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
            a.Send = (decltype(a.Send))dlsym(a.lib, "ncclSend");
            a.Recv = (decltype(a.Recv))dlsym(a.lib, "ncclRecv");
            a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
        }
        return a;
    }();
    return api;
}

// ---- device side of the exchange ----
constexpr int kExWidth = 10;       // packed fp64 words per node: 9 sums + count

// position of every local node in the canonical (sorted, unique) key list, and its statistics scattered there
static __global__ void __launch_bounds__(256) k_exchange_scatter(const uint64_t* __restrict__ key, const double* __restrict__ sums,
                                                                 const uint32_t* __restrict__ count, const uint32_t* __restrict__ first,
                                                                 uint32_t m, const uint64_t* __restrict__ canon, uint32_t C,
                                                                 double* __restrict__ packed, uint32_t* __restrict__ pfirst,
                                                                 uint32_t* __restrict__ missing) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint64_t k = key[i];
        uint32_t lo = 0, hi = C;                          // lower_bound
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (canon[mid] < k) lo = mid + 1; else hi = mid; }
        if (lo >= C || canon[lo] != k) { atomicAdd(missing, 1u); continue; }      // (cannot happen: the list is the union)
        double* o = packed + (size_t)lo * kExWidth;
#pragma unroll
        for (int j = 0; j < 9; ++j) o[j] = sums[9 * (size_t)i + j];
        o[9] = (double)count[i];
        pfirst[lo] = first[i];
    }
}

static __global__ void __launch_bounds__(256) k_exchange_init(double* __restrict__ packed, uint32_t* __restrict__ pfirst, uint32_t C) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)C * kExWidth; i += (uint64_t)gridDim.x * blockDim.x) packed[i] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) pfirst[i] = 0xFFFFFFFFu;
}

static __global__ void __launch_bounds__(256) k_exchange_unpack(const double* __restrict__ packed, uint32_t C, double* __restrict__ sums,
                                                                uint32_t* __restrict__ count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) {
        const double* p = packed + (size_t)i * kExWidth;
#pragma unroll
        for (int j = 0; j < 9; ++j) sums[9 * (size_t)i + j] = p[j];
        count[i] = (uint32_t)p[9];
    }
}

// keys padded to a common length with kEmptyKey (sorts last); after sort + unique the pad is the last entry, if present
static __global__ void k_exchange_pad(uint64_t* __restrict__ buf, uint32_t have, uint32_t padded) {
    for (uint32_t i = have + blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += gridDim.x * blockDim.x) buf[i] = kEmptyKey;
}


// ---------------------------------------------------------------------------------------------
// Owner-partitioned build (gndt_build_owned_device): points travel, statistics do not.
//
//   every rank   splits its shard by owner_of(column) into W runs of 16-B records {x, y, z, global index | weight flags}
//                (k_part_hist / k_part_scatter in owner mode; identical consecutive points already folded into weighted records)
//   all-to-all   the runs (ncclSend / ncclRecv in one group): a rank receives every point of the columns it owns
//   every rank   builds ITS columns with the ordinary PARTITION pipeline from the records (index words taken as they are):
//                statistics, labels and rows are final, nothing is merged afterwards
//   all-gather   every rank's columns as (first-seen index, node count) pairs, 8 B per column: with everybody's pairs in the
//                column-order arrays (bitmap, word weights, ncol_at) the usual prefix gives the row every local row has in the
//                map of the WHOLE cloud (k_global_rows)
// Per rank and build: (W-1)/W of its points leave (16 B each) and as many arrive, plus 8 B per column of the global map —
// against 84 B per node of the global map through a ring all-reduce for the statistics exchange above.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kNoPair = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned long long kPoisonPair = 0xFFFFFFFE00000001ull;    // first-seen index beyond any cloud: k_pairs_note counts it as bad

// ---- who owns which column: locality first ----
// A contiguous range of a scan-ordered cloud covers a patch of ground, so most points of a 32 x 32-column block (6.4 m at
// 0.2 m cells) sit on one or two ranks already.  Every rank samples kOwnerSamples points of its shard evenly and publishes
// their blocks (one fixed-size all-gather, 256 KB per rank, no host round trip); every rank then counts, for every block
// seen, the samples of each rank weighted by that rank's shard size, and gives the block to the rank that holds most of
// it — the same deterministic rule on the same data, so all ranks hold the same map.  A block with more than 1 / (4 W) of
// the whole cloud is NOT given to one rank (its columns are spread by owner_of), and so is every block no sample hit.
// Measured on the S3 terrain (8 M points, W = 2 / 4 / 8): 81 / 80 / 75 % of the points are already on their owner
// (hash ownership: 50 / 25 / 12.5 %), heaviest rank 1.01 / 1.02 / 1.08 x the mean.
constexpr uint32_t kOwnerSamples = 65536, kOwnerSlots = 1u << 20, kOwnerMsgWords = kOwnerSamples + 2, kNoSample = 0xFFFFFFFFu;
constexpr uint32_t kOwnerMapMaxRanks = 16;

// msg[0..1] = shard size, msg[2 + k] = block of the k-th sample (kNoSample: no such sample / outside the key range)
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(256) k_owner_sample(const float* __restrict__ xyz, uint64_t n, GridParams P, uint32_t* __restrict__ msg) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { msg[0] = (uint32_t)n; msg[1] = (uint32_t)(n >> 32); }
    if (k >= kOwnerSamples) return;
    uint32_t out = kNoSample;
    const uint64_t have = n < (uint64_t)kOwnerSamples ? n : (uint64_t)kOwnerSamples;
    if (k < have) {
        const uint64_t i = n < (uint64_t)kOwnerSamples ? (uint64_t)k : ((uint64_t)k * n) / kOwnerSamples;
        const float* p = xyz + i * STRIDE_FLOATS;
        int sx, sy;
        bool ok;
        column_of_point(p[0], p[1], P, sx, sy, ok);
        if (ok) out = owner_block(sx, sy);
    }
    msg[2 + k] = out;
}

// everybody's samples -> per block, the samples of every rank (bcnt[slot * W + r])
static __global__ void __launch_bounds__(256) k_owner_vote(const uint32_t* __restrict__ msgs, uint32_t W, uint32_t* __restrict__ bkey,
                                                           uint32_t* __restrict__ bcnt, uint32_t mask, uint32_t* __restrict__ full) {
    const uint64_t total = (uint64_t)W * kOwnerSamples;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(t / kOwnerSamples), k = (uint32_t)(t % kOwnerSamples);
        const uint32_t b = msgs[(size_t)r * kOwnerMsgWords + 2 + k];
        if (b == kNoSample) continue;
        const uint32_t key = b + 1u;
        uint32_t s = owner_block_slot(key) & mask;
        bool found = false;
        for (uint32_t probe = 0; probe <= mask; ++probe) {
            uint32_t cur = bkey[s];
            if (cur == 0u) { cur = atomicCAS(&bkey[s], 0u, key); if (cur == 0u) cur = key; }
            if (cur == key) { found = true; break; }
            s = (s + 1u) & mask;
        }
        if (found) atomicAdd(&bcnt[(size_t)s * W + r], 1u); else atomicAdd(full, 1u);
    }
}

// per block: the rank that holds most of it (samples x shard size; ties: the lower rank), unless the block is too hot
static __global__ void __launch_bounds__(256) k_owner_pick(const uint32_t* __restrict__ msgs, uint32_t W, const uint32_t* __restrict__ bkey,
                                                           const uint32_t* __restrict__ bcnt, uint32_t slots, uint8_t* __restrict__ bown) {
    unsigned long long cloud = 0;
    for (uint32_t r = 0; r < W; ++r) cloud += (unsigned long long)msgs[(size_t)r * kOwnerMsgWords] | ((unsigned long long)msgs[(size_t)r * kOwnerMsgWords + 1] << 32);
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) {
        if (bkey[s] == 0u) continue;
        unsigned long long best = 0, sum = 0;
        uint32_t who = 0xFFu;
        for (uint32_t r = 0; r < W; ++r) {
            const unsigned long long n_r = (unsigned long long)msgs[(size_t)r * kOwnerMsgWords] | ((unsigned long long)msgs[(size_t)r * kOwnerMsgWords + 1] << 32);
            const unsigned long long w = (unsigned long long)bcnt[(size_t)s * W + r] * n_r;     // < 2^16 * 2^31
            sum += w;
            if (w > best) { best = w; who = r; }
        }
        // share of the cloud = sum / (samples per rank * cloud): too hot for ONE rank above 1 / (4 W)
        if (sum * 4ull * W > (unsigned long long)kOwnerSamples * cloud) who = 0xFFu;
        bown[s] = (uint8_t)who;
    }
}

// the columns of the local map: (first-seen index of the column, its node count), in any order.  A workgroup takes a chunk
// of kColChunk rows: it counts the chunk's column heads, reserves their places with ONE memory-side atomic (a per-wave
// atomic on the single counter word serialises: 26 k of them cost 0.3 ms on a 1.7 M-row map) and writes them out.
constexpr uint32_t kColChunk = 4096;
static __global__ void __launch_bounds__(256) k_owned_columns(const uint32_t* __restrict__ first_idx, const uint32_t* __restrict__ row_ncol,
                                                              const Counters* __restrict__ cnt, const PartCounters* __restrict__ pc,
                                                              unsigned long long* __restrict__ pairs,
                                                              uint32_t pairs_cap, uint32_t* __restrict__ n_pairs,
                                                              uint32_t* __restrict__ row_of_pair /* nullable: the column's first row */) {
    if (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) return;   // the build is re-run: no rows yet
    __shared__ uint32_t heads, base, cursor;
    const uint32_t n = cnt->num_nodes;
    const int lane = threadIdx.x & 63;
    for (uint32_t c0 = blockIdx.x * kColChunk; c0 < n; c0 += gridDim.x * kColChunk) {
        if (threadIdx.x == 0) { heads = 0; cursor = 0; }
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t r = c0 + threadIdx.x; r < min(c0 + kColChunk, n); r += 256) mine += row_ncol[r] != 0u ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_down((int)mine, off, 64);
        if (lane == 0 && mine) atomicAdd(&heads, mine);
        __syncthreads();
        if (threadIdx.x == 0) base = heads ? atomicAdd(n_pairs, heads) : 0u;
        __syncthreads();
        for (uint32_t r0 = c0; r0 < min(c0 + kColChunk, n); r0 += 256) {          // (uniform trip count: whole waves reach the ballot)
            const uint32_t r = r0 + threadIdx.x;
            const uint32_t nc = r < n && r < c0 + kColChunk ? row_ncol[r] : 0u;
            const unsigned long long m = __ballot(nc != 0u);
            if (!m) continue;
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&cursor, (uint32_t)__popcll(m));
            wbase = (uint32_t)__shfl((int)wbase, 0, 64);
            if (nc) {
                const uint32_t pos = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < pairs_cap) { pairs[pos] = ((unsigned long long)first_idx[r] << 32) | nc; if (row_of_pair) row_of_pair[pos] = r; }
            }
        }
        __syncthreads();
    }
}

// what a rank tells the others before the pairs travel: how many it has, and whether its build has to be re-run first
// (and its share of the whole map's totals, so that no further collective is needed for them)
// [4]: a status code — non-zero means this rank cannot go on (its error code, include/gndt.h): every rank reads everybody's
// word after the all-gather and they all leave the sequence together (nobody is left waiting in the next collective);
// [5]: rows of this rank's map (what gndt_gather_owned_map_device moves).
// [6], [7]: the capacities (in pairs) of this rank's pair send buffer and of its buffer for everybody's pairs: with them every rank
// knows whether ANY rank has to grow a buffer before the pairs travel, and only then do the ranks spend an agreement round on it.
constexpr int kColMsgWords = 8;
static __global__ void k_owned_status(const uint32_t* __restrict__ n_pairs, const PartCounters* __restrict__ pc, const Counters* __restrict__ cnt,
                                      unsigned long long owned_points, unsigned long long* __restrict__ msg, unsigned long long pairs_cap,
                                      unsigned long long pairs_all_cap) {
    msg[6] = pairs_cap;
    msg[7] = pairs_all_cap;
    msg[0] = *n_pairs;
    msg[1] = (pc->lds_overflow | pc->stage_overflow | pc->index_overflow | pc->part_overflow) ? 1ull : 0ull;
    msg[2] = cnt->num_slopes;
    msg[3] = owned_points;
    msg[4] = 0ull;
    msg[5] = cnt->num_nodes;
}
// the status word that travels with a rank's row of the W x W send-count matrix: the host's code if it has one, else what the
// split found on the device (points outside the key range)
static __global__ void k_split_status(const Counters* __restrict__ split_cnt, uint32_t host_code, uint32_t key_range_code, uint32_t* __restrict__ word,
                                      uint32_t receive_capacity) {
    word[0] = host_code ? host_code : ((split_cnt && split_cnt->err_key_range) ? key_range_code : 0u);
    word[1] = receive_capacity;      // (records this rank can take in the exchange without growing its buffer)
}

static __global__ void __launch_bounds__(256) k_pairs_pad(unsigned long long* __restrict__ pairs, uint32_t have, uint32_t padded) {
    for (uint32_t i = have + blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += gridDim.x * blockDim.x) pairs[i] = kNoPair;
}

// Everybody's columns into the column order of the WHOLE map.  A column costs ONE memory-side atomic here (they retire at
// ~20 G/s chip-wide, and S3 has 4 M columns): bit and weight of its bitmap word are the two halves of one 64-bit word (a bit
// is set once, so add = or), split into the arrays the prefix and the row kernels read afterwards.
static __global__ void __launch_bounds__(256) k_order_clear(unsigned long long* __restrict__ gw, uint64_t words,
                                                            unsigned long long* __restrict__ totals, uint32_t* __restrict__ bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) gw[i] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) { totals[0] = 0ull; totals[1] = 0ull; *bad = 0u; }
}

// node and column totals of the whole map are summed per workgroup first (one pair of memory-side atomics per workgroup)
static __global__ void __launch_bounds__(256) k_pairs_note(const unsigned long long* __restrict__ pairs, uint64_t n,
                                                           unsigned long long* __restrict__ gw, uint32_t* __restrict__ ncol_at,
                                                           uint64_t words, unsigned long long* __restrict__ totals, uint32_t* __restrict__ bad) {
    __shared__ unsigned long long s_nodes, s_cols;
    if (threadIdx.x == 0) { s_nodes = 0ull; s_cols = 0ull; }
    __syncthreads();
    unsigned long long nodes = 0, cols = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long pr = pairs[i];
        if (pr == kNoPair) continue;
        const uint32_t cf = (uint32_t)(pr >> 32), nc = (uint32_t)pr;
        if ((uint64_t)(cf >> 5) >= words) { atomicAdd(bad, 1u); continue; }
        atomicAdd(&gw[cf >> 5], ((unsigned long long)nc << 32) | (1ull << (cf & 31u)));
        ncol_at[cf] = nc;
        nodes += nc; ++cols;
    }
    nodes = (unsigned long long)wave_sum((double)nodes);      // (exact: far below 2^53)
    cols = (unsigned long long)wave_sum((double)cols);
    if ((threadIdx.x & 63) == 0 && cols) { atomicAdd(&s_nodes, nodes); atomicAdd(&s_cols, cols); }
    __syncthreads();
    if (threadIdx.x == 0 && s_cols) { atomicAdd(&totals[0], s_nodes); atomicAdd(&totals[1], s_cols); }
}

// ---- the same, sliced: rank t orders only the columns whose first-seen index falls into ITS slice of the index range ----
// With every rank noting all columns the ordering costs W times the memory-side atomics of the single-GPU build (S3: 4 M
// columns; profiles/r02_owner_threads.json: 0.8 -> 2.2 ms from 2 to 8 ranks).  Here rank t owns the bitmap words
// [t * slice_words, (t + 1) * slice_words): it notes the pairs of that slice (1 / W of the atomics), takes the prefix inside
// it, and writes for every pair of the slice — whoever owns the column — the place of the column RELATIVE to the slice.
// A sum reduce-scatter over the ranks hands every rank the places of its own pairs (each is non-zero on one rank at most), an
// all-gather of the W slice totals turns them into rows of the whole map.
static __global__ void __launch_bounds__(256) k_pairs_note_slice(const unsigned long long* __restrict__ pairs, uint64_t n, uint64_t words,
                                                                 uint64_t lo_word, uint64_t hi_word, unsigned long long* __restrict__ gw,
                                                                 uint32_t* __restrict__ ncol_at, unsigned long long* __restrict__ slice_tot,
                                                                 uint32_t* __restrict__ bad) {
    __shared__ unsigned long long s_nodes, s_cols;
    if (threadIdx.x == 0) { s_nodes = 0ull; s_cols = 0ull; }
    __syncthreads();
    unsigned long long nodes = 0, cols = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long pr = pairs[i];
        if (pr == kNoPair) continue;
        const uint32_t cf = (uint32_t)(pr >> 32), nc = (uint32_t)pr;
        const uint64_t w = cf >> 5;
        if (w >= words) { atomicAdd(bad, 1u); continue; }         // (every rank sees every pair: they all count it)
        if (w < lo_word || w >= hi_word) continue;
        atomicAdd(&gw[w - lo_word], ((unsigned long long)nc << 32) | (1ull << (cf & 31u)));
        ncol_at[cf] = nc;
        nodes += nc; ++cols;
    }
    nodes = (unsigned long long)wave_sum((double)nodes);
    cols = (unsigned long long)wave_sum((double)cols);
    if ((threadIdx.x & 63) == 0 && cols) { atomicAdd(&s_nodes, nodes); atomicAdd(&s_cols, cols); }
    __syncthreads();
    if (threadIdx.x == 0 && s_cols) { atomicAdd(&slice_tot[0], s_nodes); atomicAdd(&slice_tot[1], s_cols); }
}
// place of every pair of the slice relative to the slice's first row (0 for the pairs of other slices)
static __global__ void __launch_bounds__(256) k_pair_places(const unsigned long long* __restrict__ pairs, uint64_t n, uint64_t lo_word, uint64_t hi_word,
                                                            const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_base,
                                                            const uint32_t* __restrict__ ncol_at, uint32_t* __restrict__ place) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long pr = pairs[i];
        uint32_t out = 0u;
        if (pr != kNoPair) {
            const uint32_t cf = (uint32_t)(pr >> 32);
            const uint64_t w = cf >> 5;
            if (w >= lo_word && w < hi_word) {
                uint32_t m = bitmap[w - lo_word] & ((1u << (cf & 31u)) - 1u);
                out = word_base[w - lo_word];
                while (m) { out += ncol_at[((uint32_t)w << 5) + (uint32_t)__builtin_ctz(m)]; m &= m - 1u; }
            }
        }
        place[i] = out;
    }
}
// everybody's slice totals {nodes, columns} -> the first row of every slice (slice_row[t]) and the totals of the whole map
static __global__ void k_slice_rows(const unsigned long long* __restrict__ slice_tot_all, uint32_t W, unsigned long long* __restrict__ slice_row,
                                    unsigned long long* __restrict__ totals) {
    unsigned long long nodes = 0, cols = 0;
    for (uint32_t t = 0; t < W; ++t) { slice_row[t] = nodes; nodes += slice_tot_all[2 * t]; cols += slice_tot_all[2 * t + 1]; }
    totals[0] = nodes; totals[1] = cols;
}
// this rank's pairs, their places and the slices' first rows -> the row of every local row in the map of the whole cloud
static __global__ void __launch_bounds__(256) k_global_rows_sliced(const unsigned long long* __restrict__ pairs, const uint32_t* __restrict__ row_of_pair,
                                                                   const uint32_t* __restrict__ place, uint32_t n_pairs, uint64_t slice_words,
                                                                   const unsigned long long* __restrict__ slice_row, uint32_t* __restrict__ global_row) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_pairs; j += gridDim.x * blockDim.x) {
        const unsigned long long pr = pairs[j];
        if (pr == kNoPair) continue;
        const uint32_t cf = (uint32_t)(pr >> 32), nc = (uint32_t)pr, r = row_of_pair[j];
        const uint32_t row = (uint32_t)slice_row[(cf >> 5) / slice_words] + place[j];
        for (uint32_t i = 0; i < nc; ++i) global_row[r + i] = row + i;
    }
}
static __global__ void __launch_bounds__(256) k_add_f64(double* __restrict__ dst, const double* __restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
static __global__ void __launch_bounds__(256) k_min_u32(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = min(dst[i], src[i]);
}
static __global__ void __launch_bounds__(256) k_add_u32(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

static __global__ void __launch_bounds__(256) k_order_split(const unsigned long long* __restrict__ gw, uint64_t words,
                                                            uint32_t* __restrict__ bitmap, uint32_t* __restrict__ word_weight) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = gw[i];
        bitmap[i] = (uint32_t)v; word_weight[i] = (uint32_t)(v >> 32);
    }
}

// row of every local row in the map of the whole cloud: the column's position from the global column order, then the nodes
// of the column in their local (= global) order.  One lane per column head; columns are short.
static __global__ void __launch_bounds__(256) k_global_rows(const uint32_t* __restrict__ first_idx, const uint32_t* __restrict__ row_ncol,
                                                            const Counters* __restrict__ cnt, const uint32_t* __restrict__ bitmap,
                                                            const uint32_t* __restrict__ word_base, const uint32_t* __restrict__ ncol_at,
                                                            uint32_t* __restrict__ global_row) {
    const uint32_t n = cnt->num_nodes;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint32_t nc = row_ncol[r];
        if (!nc) continue;
        const uint32_t cf = first_idx[r], w = cf >> 5;
        uint32_t m = bitmap[w] & ((1u << (cf & 31u)) - 1u);
        uint32_t row = word_base[w];
        while (m) { row += ncol_at[(w << 5) + (uint32_t)__builtin_ctz(m)]; m &= m - 1u; }
        for (uint32_t i = 0; i < nc; ++i) global_row[r + i] = row + i;
    }
}

// ---------------------------------------------------------------------------------------------
// The assembled map (gndt_gather_owned_map_device): a rank's result rows travel as packed records of kPackedRowWords 32-bit
// words — the 76-byte row of gndt_cells (sx, sy, sz, count, first_idx, mean[3], cov[6], rough, normal[3], flags: 19 words),
// the column index the cost map reads (row_ncol) and the row's place in the map of the whole cloud — and are scattered by
// that place into the result arrays of the handle that adopts them.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kPackedRowWords = 21, kNoRow = 0xFFFFFFFFu;
static __global__ void __launch_bounds__(256) k_rows_pack(OutView o, const uint32_t* __restrict__ row_ncol, const uint32_t* __restrict__ global_row,
                                                          uint32_t n, uint32_t padded, uint32_t* __restrict__ rows) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < padded; r += gridDim.x * blockDim.x) {
        uint32_t* w = rows + (size_t)r * kPackedRowWords;
        if (r >= n) { w[20] = kNoRow; continue; }                   // (padding of a fixed-size all-gather)
        w[0] = (uint32_t)o.sx[r]; w[1] = (uint32_t)o.sy[r]; w[2] = (uint32_t)o.sz[r]; w[3] = o.count[r]; w[4] = o.first_idx[r];
        for (int k = 0; k < 3; ++k) w[5 + k] = __float_as_uint(o.mean[3 * (size_t)r + k]);
        for (int k = 0; k < 6; ++k) w[8 + k] = __float_as_uint(o.cov[6 * (size_t)r + k]);
        w[14] = __float_as_uint(o.rough[r]);
        for (int k = 0; k < 3; ++k) w[15 + k] = __float_as_uint(o.normal[3 * (size_t)r + k]);
        w[18] = o.flags[r]; w[19] = row_ncol[r]; w[20] = global_row[r];
    }
}
// tally[0] = rows written, tally[1] = rows whose place lies beyond the map
static __global__ void __launch_bounds__(256) k_rows_adopt(const uint32_t* __restrict__ rows, uint64_t n, uint32_t total, OutView o,
                                                           uint32_t* __restrict__ row_ncol, uint32_t* __restrict__ tally) {
    uint32_t wrote = 0, bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t* w = rows + i * kPackedRowWords;
        const uint32_t r = w[20];
        if (r == kNoRow) continue;
        if (r >= total) { ++bad; continue; }
        o.sx[r] = (int32_t)w[0]; o.sy[r] = (int32_t)w[1]; o.sz[r] = (int32_t)w[2]; o.count[r] = w[3]; o.first_idx[r] = w[4];
        for (int k = 0; k < 3; ++k) o.mean[3 * (size_t)r + k] = __uint_as_float(w[5 + k]);
        for (int k = 0; k < 6; ++k) o.cov[6 * (size_t)r + k] = __uint_as_float(w[8 + k]);
        o.rough[r] = __uint_as_float(w[14]);
        for (int k = 0; k < 3; ++k) o.normal[3 * (size_t)r + k] = __uint_as_float(w[15 + k]);
        o.flags[r] = w[18]; row_ncol[r] = w[19];
        ++wrote;
    }
    for (int off = 32; off > 0; off >>= 1) { wrote += (uint32_t)__shfl_down((int)wrote, off, 64); bad += (uint32_t)__shfl_down((int)bad, off, 64); }
    if ((threadIdx.x & 63) == 0) { if (wrote) atomicAdd(&tally[0], wrote); if (bad) atomicAdd(&tally[1], bad); }
}

}  // namespace gndt
