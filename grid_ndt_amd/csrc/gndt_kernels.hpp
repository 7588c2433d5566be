// gndt_kernels.hpp — strategy ATOMIC: the persistent node table in HBM (gfx950 / MI355X).
//
// Data layout (owned by the handle, gndt_api.hip):
//   keys[cap]        u64   open-addressing node table, kEmptyKey = free; linear probing
//   acc[cap]         NodeAcc (80 B): additive cell-local statistics of the node in that slot
//   node_slot[C]     u32   the occupied slots in insertion order: a node is appended when its key is inserted,
//                          so the list is always valid and nothing ever scans the whole table
//   col_keys / col_first / col_cnt / col_head [cap]   column table, rebuilt for the touched columns by finalize
//   node_next[C], col_slot_of_node[C]                 per-column linked list of nodes, column slot of each node
//   aux[cap]         fp32 mean-z of the node (for the slope test)
//
// Kernels here: k_accumulate (points -> statistics; src/receiver.cpp:41-93), the table clears, and the
// statistics export / merge used by table growth and by the multi-GPU exchange.  Finalisation (labels, order,
// result rows) is in gndt_table.hpp and reuses the staging rows and ordering kernels of the partition path.
// Every size a kernel needs is read from device memory (Counters), so accumulate + finalize never wait for the
// host: an incremental update can be captured in a hipGraph and replayed per frame (BASELINE configs[3]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gndt_math.hpp"

namespace gndt {

struct NodeAcc {          // 80 bytes
    double s[9];          // Sum v (3), Sum v v^T upper triangle xx,xy,xz,yy,yz,zz (6)
    uint32_t count;
    uint32_t first;       // min point index; 0xFFFFFFFF when empty
};
static_assert(sizeof(NodeAcc) == 80, "NodeAcc layout");

struct SlotAux {
    float mean_z;         // fp32 centroid z if count >= min_points else 0 (OcNode::xyz_centroid stays zero)
    uint32_t flags;       // GNDT_FLAG_*
};

struct Counters {
    uint32_t num_nodes;       // nodes in the table (ATOMIC) / staged rows (PARTITION)
    uint32_t num_columns;
    uint32_t num_slopes;
    uint32_t err_key_range;   // points outside the key range
    uint32_t err_table_full;  // inserts that found no slot
    uint32_t stream_pos;      // points accumulated so far by gndt_update* (device-side first_idx base)
    uint32_t prev_nodes;      // nodes that went through the last finalize (their column entries exist)
    uint32_t epoch;           // stamps "touched since the last finalize" marks; bumped by every finalize, never 0 once in use
    uint32_t n_touched;       // nodes an incremental update has touched so far (list: touched[])
    uint32_t n_tcols;         // columns holding a touched node (list: touched_cols[])
    uint32_t n_work;          // nodes of those columns (list: reuses touched[])
    uint32_t n_dead;          // gndt_remove*: nodes whose last point was taken away (dropped by the compaction that follows)
    uint32_t err_remove;      // gndt_remove*: points whose node does not exist or is already empty
    uint32_t first_word;      // incremental finalisation: first bitmap word whose column order changed in this frame (a column
                              //   gained a node or is new); rows of columns in front of it keep their places.  0xFFFFFFFF: none
    uint32_t ticket;          // k_clear_used: workgroups that are done (the last one zeroes the counters); 0 between kernels
    uint32_t table_gen;       // which allocation of the node table the counters belong to (set by k_clear_all: once per allocation).
                              //   A reset recorded into a hipGraph carries the generation of ITS day: replayed after the table has
                              //   been reallocated it must not touch anything (its pointers are the old table's) ...
    uint32_t table_unclean;   //   ... and cannot clear the table that took its place: it says so here, and the next reset that does
                              //   hold the current table clears ALL of it instead of the listed slots (round 5, tools/fuzz_graph.py
                              //   with eager table-path builds between the replays: a GPU memory fault — the recorded reset walked
                              //   the old table's node list with the new table's node count)
    uint32_t part_owned;      // 1 while num_nodes counts a PARTITION build's staged rows (set by k_part_clear) and NOT the node list of
                              //   the table, which is empty then (every partition build resets a dirty table first; a captured one
                              //   always records the reset): k_clear_used must not walk node_slot[0 .. num_nodes) — past the list the
                              //   slots are whatever the allocation held (round 4, tools/fuzz_graph.py --seed 6: the second replay of a
                              //   PARTITION build captured after an ATOMIC fallback cleared "slots" read from there — a GPU memory fault)
};

// Spatially BLOCKED buckets (round 6, gndt_blocked.hpp): a bucket is a block of 2^shx x 2^shy columns x 2^shz levels = 512 nodes of the
// occupied key box, so a node's place in the bucket's table is computed from its key — no index, no search.  cx = sx > 0 ? sx - 1 : sx
// (likewise cy, cz) makes the signed indices contiguous (there is no index 0).  on = 0: buckets by column hash (everything else).
struct BlockMap {
    int on;
    int x0, y0, z0;            // smallest contiguous index of the box on every axis
    int shx, shy, shz;         // log2 of the block's extent; shx + shy + shz = 9
    int nx, ny;                // blocks along x and y (one layer of blocks: the box is at most 2^shz levels high); buckets = nx * ny
};
struct GridParams {
    float ox, oy, oz;
    float grid_len, z_len, slope_interval;
    int demand, min_points;
    float inv_grid, inv_z;     // RN(1 / grid_len), RN(1 / z_len): axis_index_fast (gndt_math.hpp)
    BlockMap blk;
};

struct OutView {
    int32_t *sx, *sy, *sz;
    uint32_t *count, *first_idx;
    float *mean, *cov, *rough, *normal;
    uint32_t* flags;
};

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// table helpers
// ---------------------------------------------------------------------------------------------
// Find the slot holding `key`, inserting it if absent.  Returns cap on failure (table full).
// A plain load may return a stale EMPTY (per-XCD L2s are not coherent) but never a stale key,
// because a slot's key is written once; the CAS executes at the memory side and settles it.
__device__ __forceinline__ uint32_t find_or_insert(uint64_t* __restrict__ keys, uint32_t cap_mask, uint64_t key,
                                                   bool& inserted) {
    inserted = false;
    uint32_t slot = (uint32_t)mix64(key) & cap_mask;
    for (uint32_t probe = 0; probe <= cap_mask; ++probe) {
        uint64_t k = keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) {
            unsigned long long old = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                                               (unsigned long long)key);
            if (old == kEmptyKey) { inserted = true; return slot; }
            if (old == key) return slot;
        }
        slot = (slot + 1) & cap_mask;
    }
    return cap_mask + 1;
}

// Append the slots of freshly inserted nodes to the node list: one counter atomic per wave instruction (the
// counter is ONE word; per-lane atomics on it would serialise at the memory side).
__device__ __forceinline__ void append_new_nodes(bool inserted, uint32_t slot, uint32_t* __restrict__ node_slot,
                                                 uint32_t* __restrict__ index_of_slot, Counters* __restrict__ cnt) {
    const unsigned long long m = __ballot(inserted);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = (int)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cnt->num_nodes, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (inserted) {
        const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        node_slot[idx] = slot;
        index_of_slot[slot] = idx;
    }
}

// Incremental updates: the first point that reaches a node since the last finalize puts the node's slot on the touched
// list (wave-aggregated append: the list counter is ONE word).
__device__ __forceinline__ void mark_touched(uint32_t slot, uint32_t epoch, uint32_t* __restrict__ touch_epoch,
                                             uint32_t* __restrict__ touched, Counters* __restrict__ cnt) {
    const bool first = atomicExch(&touch_epoch[slot], epoch) != epoch;
    const unsigned long long m = __ballot(first);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = (int)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cnt->n_touched, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (first) touched[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = slot;
}

// Lookup only.  Returns cap_mask+1 when absent.
__device__ __forceinline__ uint32_t find_slot(const uint64_t* __restrict__ keys, uint32_t cap_mask, uint64_t key) {
    uint32_t slot = (uint32_t)mix64(key) & cap_mask;
    for (uint32_t probe = 0; probe <= cap_mask; ++probe) {
        uint64_t k = keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) return cap_mask + 1;
        slot = (slot + 1) & cap_mask;
    }
    return cap_mask + 1;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// k_clear_all: whole-table initialisation (once at create / after growth)
// ---------------------------------------------------------------------------------------------
static __global__ void k_clear_all(uint64_t* keys, NodeAcc* acc, uint64_t* col_keys, uint32_t* col_first, uint32_t* col_cnt,
                            uint32_t* col_head, uint32_t* touch_epoch, uint32_t* col_epoch, uint32_t cap, Counters* cur, uint32_t gen) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { cur->table_gen = gen; cur->table_unclean = 0u; }      // (a new table: this generation, clean)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        keys[i] = kEmptyKey;
        col_keys[i] = kEmptyKey;
        col_first[i] = 0xFFFFFFFFu;
        col_cnt[i] = 0u;
        col_head[i] = 0xFFFFFFFFu;
        touch_epoch[i] = 0u;
        col_epoch[i] = 0u;
        NodeAcc a;
        for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
        a.count = 0; a.first = 0xFFFFFFFFu;
        acc[i] = a;
    }
}

// k_clear_used: empty the map by visiting only its occupied slots (O(C), not O(cap)).  Nodes that went through a
// finalize (i < prev_nodes) also own a column-table entry.
// `table_cleared`: the caller has emptied the CURRENT node table (or it was empty).  If it has not, and the counters describe a
// table that holds nodes, the table stays behind unclean (see Counters::table_unclean).
static __device__ __forceinline__ void zero_counters(Counters* c, bool table_cleared) {
    if (!table_cleared && c->part_owned == 0u && c->num_nodes != 0u) c->table_unclean = 1u;
    c->num_nodes = 0; c->num_columns = 0; c->num_slopes = 0; c->err_key_range = 0; c->err_table_full = 0;
    c->stream_pos = 0; c->prev_nodes = 0; c->n_touched = 0; c->n_tcols = 0; c->n_work = 0; c->n_dead = 0; c->err_remove = 0;
    c->epoch = c->epoch + 1u;                     // stale touch marks of the previous map can never match again
    c->part_owned = 0u;                           // (num_nodes is the table's node list again, empty)
}

static __global__ void k_clear_used(uint64_t* keys, NodeAcc* acc, uint64_t* col_keys, uint32_t* col_first, uint32_t* col_cnt,
                             uint32_t* col_head, const uint32_t* node_slot, const uint32_t* col_slot_of_node,
                             Counters* cur, uint32_t cap, uint32_t gen) {
    // mine: these pointers are the handle's CURRENT table.  Else the reset was recorded for a table that has been reallocated since
    // (the old arrays are retired, not freed: Handle::retired): the counters then describe the new table, not this one — nothing of
    // them may be used as an index here.  The old table is cleared WHOLE instead, so that the recorded build that follows stays
    // consistent within its own (old) buffers (with the nodes of the replay before left in it, its column lists ran in circles:
    // a hang; the host reports such a replay as stale, but it must end).
    const bool mine = cur->table_gen == gen;
    const bool everything = !mine || cur->table_unclean != 0u;     // (unclean: a stale reset could not clear the current table: all of it, now)
    const bool listed = mine && !everything && cur->part_owned == 0u;    // (part_owned: the counters are a PARTITION build's and the table is empty)
    const uint32_t n = listed ? min(cur->num_nodes, cap) : 0u, np = listed ? cur->prev_nodes : 0u;
    if (everything) {
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
            keys[i] = kEmptyKey;
            col_keys[i] = kEmptyKey;
            col_first[i] = 0xFFFFFFFFu;
            col_cnt[i] = 0u;
            col_head[i] = 0xFFFFFFFFu;
            NodeAcc a;
            for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
            a.count = 0; a.first = 0xFFFFFFFFu;
            acc[i] = a;
        }
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = node_slot[i];
        if (slot >= cap) continue;
        keys[slot] = kEmptyKey;
        NodeAcc a;
        for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
        a.count = 0; a.first = 0xFFFFFFFFu;
        acc[slot] = a;
        if (i < np) {
            const uint32_t cs = col_slot_of_node[i];   // nodes of one column write the same values: benign
            if (cs >= cap) continue;
            col_keys[cs] = kEmptyKey;
            col_first[cs] = 0xFFFFFFFFu;
            col_cnt[cs] = 0u;
            col_head[cs] = 0xFFFFFFFFu;
        }
    }
    // The counters go to zero in the same launch (a one-thread kernel behind this one cost a small build ~5 us): every thread
    // has read them above; the workgroup that takes the last ticket knows that all the others have, too.
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&cur->ticket, 1u) == gridDim.x - 1u) {
        zero_counters(cur, mine);
        if (mine && everything) cur->table_unclean = 0u;
        cur->ticket = 0u;
    }
}

// all counters to zero (after the clear that read them)
static __global__ void k_zero_counters(Counters* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) zero_counters(c, false);      // (no table cleared: if the counters describe one that holds nodes, it is marked)
}

// A record's index word.  Bit 31: the record stands for 64 identical consecutive points (one wave of the partition pass:
// the (0,0,0) padding of the reference's own clouds); with bit 30 as well for 512 of them (all eight groups a
// wave handles in a tile).  Weighted records keep the index of their FIRST point in the low 30 bits; clouds of 2^30 points
// or more are simply not compressed.
constexpr uint32_t kWeight64Flag = 0x80000000u, kWeight512Flag = 0x40000000u, kWeightIndexLimit = 0x40000000u;
__host__ __device__ __forceinline__ uint32_t record_weight(uint32_t iw) { return (iw & kWeight64Flag) ? ((iw & kWeight512Flag) ? 512u : 64u) : 1u; }
__host__ __device__ __forceinline__ uint32_t record_index(uint32_t iw) { return (iw & kWeight64Flag) ? (iw & (kWeightIndexLimit - 1u)) : iw; }

// ---------------------------------------------------------------------------------------------
// k_accumulate (strategy ATOMIC): one point per thread-iteration
//   uniformDivision (src/receiver.cpp:41-93) + transMortonXYZ (include/map2D.h:950-976), with the
//   per-node point list replaced by additive statistics.
// HBM roofline view: reads 12 (or 16) B/point; the 11 atomics per point execute at the memory side.
// When every lane of a wave holds the same key (the (0,0,0) padding of the reference's own clouds,
// SURVEY §4) the wave reduces in registers and issues one set of atomics.
// first_idx of point i = first_base + i, or cnt->stream_pos + i when `base_from_device` (incremental updates:
// the base then lives on the device, so a captured graph can be replayed frame after frame).
// ---------------------------------------------------------------------------------------------
// REC: the input is 16-B records {x, y, z, index word} (the owner-partitioned build of a sharded cloud, when the records do not
// fit the partition pipeline): the point index and the weight come from the word.
template <int STRIDE_FLOATS, bool REC = false>
__global__ void __launch_bounds__(kBlock) k_accumulate(const float* __restrict__ xyz, uint64_t n, uint32_t first_base,
                                                       int base_from_device, GridParams P, uint64_t* __restrict__ keys,
                                                       NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                                       uint32_t* __restrict__ node_slot, uint32_t* __restrict__ index_of_slot,
                                                       uint32_t* __restrict__ touch_epoch, uint32_t* __restrict__ touched,
                                                       int mark, Counters* __restrict__ cnt) {
    __shared__ double s_q[kBlock / 64][64 * 9];          // per wave: the leading lanes' nine sums, transposed for the atomics
    __shared__ uint32_t s_slot[kBlock / 64][64];
    __shared__ uint32_t s_app[2][kBlock / 64 + 1];       // list appends of the block: per wave counts, then the block's base
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t epoch = cnt->epoch;
    const uint64_t n_round = (n + kBlock - 1) / kBlock * kBlock;   // whole BLOCKS: the waves of a block stay together (block barriers below)
    const uint32_t base = base_from_device ? cnt->stream_pos : first_base;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool live = i < n;
        static_assert(!REC || STRIDE_FLOATS == 4, "records are 16 bytes");
        float px = 0.f, py = 0.f, pz = 0.f;
        uint32_t word = 0u;
        if (live) {
            const float* p = xyz + i * STRIDE_FLOATS;
            px = p[0]; py = p[1]; pz = p[2];
            if constexpr (REC) word = __float_as_uint(p[3]);
        }
        const uint32_t wn = REC ? record_weight(word) : 1u;       // (64 or 512 identical points in one record: exact, powers of two)
        PointKey k = point_key(px, py, pz, P.ox, P.oy, P.oz, P.grid_len, P.z_len);
        bool ok = live && k.ok;
        if (live && !k.ok) atomicAdd(&cnt->err_key_range, 1u);
        uint64_t key = ok ? pack_key(k.sx, k.sy, k.sz) : kEmptyKey;
        double v[3] = {0, 0, 0};
        if (ok) {
            v[0] = (double)px - axis_centre(k.sx, P.ox, P.grid_len);
            v[1] = (double)py - axis_centre(k.sy, P.oy, P.grid_len);
            v[2] = (double)pz - axis_centre(k.sz, P.oz, P.z_len);
        }
        double q[9] = {v[0], v[1], v[2], v[0] * v[0], v[0] * v[1], v[0] * v[2], v[1] * v[1], v[1] * v[2], v[2] * v[2]};
        if constexpr (REC) {
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] *= (double)wn;
        }
        uint32_t pidx = REC ? record_index(word) : base + (uint32_t)i;

        // wave-uniform key?  (all 64 lanes live, ok and equal)
        const uint64_t key0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(key >> 32)) << 32) |
                              (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)key);
        const bool uniform = __all(ok && key == key0);
        // what this lane appends to the lists at the end of the iteration: a node it created, a node it is the first to touch
        uint32_t app_slot = 0;
        bool app_new = false, app_touch = false;
        if (uniform) {
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] = wave_sum(q[j]);
            uint32_t pmin = pidx, wsum = wn;
            for (int off = 32; off > 0; off >>= 1) {
                pmin = min(pmin, (uint32_t)__shfl_down((int)pmin, off, 64));
                if constexpr (REC) wsum += (uint32_t)__shfl_down((int)wsum, off, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                bool inserted;
                uint32_t slot = find_or_insert(keys, cap_mask, key, inserted);
                if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); }
                else {
                    app_slot = slot; app_new = inserted;
                    app_touch = mark && atomicExch(&touch_epoch[slot], epoch) != epoch;
                    NodeAcc* a = acc + slot;
#pragma unroll
                    for (int j = 0; j < 9; ++j) unsafeAtomicAdd(&a->s[j], q[j]);
                    atomicAdd(&a->count, REC ? wsum : 64u);
                    atomicMin(&a->first, pmin);
                }
            }
        } else {
            // The lanes of a wave that hold the same node are added as ONE contribution: the wave sorts its keys (a bitonic
            // network over (key, lane) in registers), gathers the contributions into that order and reduces every run of equal
            // keys onto its first lane.  A spinning LiDAR stays in a 0.2 m voxel for a few azimuth steps and comes back to it
            // after a step next door: the 64 points of a wave of a 131 k-point terrain frame fall into 26 nodes (36 runs before
            // sorting), and every node saved is 11 memory-side atomics and a table probe — which is all this kernel waits for;
            // the ~600 extra vector instructions per wave are free.  (Order of the fp64 additions: as unordered as before.)
            const int lane = threadIdx.x & 63;
            uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32), src = (uint32_t)lane;
#pragma unroll
            for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
                for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                    const uint32_t olo = (uint32_t)__shfl_xor((int)klo, jj, 64), ohi = (uint32_t)__shfl_xor((int)khi, jj, 64),
                                   osrc = (uint32_t)__shfl_xor((int)src, jj, 64);
                    const bool mine_less = khi < ohi || (khi == ohi && (klo < olo || (klo == olo && src < osrc)));
                    const bool keep_min = ((lane & jj) == 0) == ((lane & kk) == 0);
                    if (keep_min != mine_less) { klo = olo; khi = ohi; src = osrc; }
                }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] = __shfl(q[j], (int)src, 64);
            pidx = (uint32_t)__shfl((int)pidx, (int)src, 64);
            ok = __shfl((int)ok, (int)src, 64) != 0;
            key = ((uint64_t)khi << 32) | (uint64_t)klo;
            const uint32_t plo = (uint32_t)__shfl_up((int)klo, 1, 64), phi = (uint32_t)__shfl_up((int)khi, 1, 64);
            const bool head = lane == 0 || plo != klo || phi != khi;
            const unsigned long long hm = __ballot(head);
            const uint32_t rid = (uint32_t)__popcll(hm & (~0ull >> (63 - lane)));          // runs that start at or before this lane
            uint32_t wn_s = wn;
            if constexpr (REC) wn_s = (uint32_t)__shfl((int)wn, (int)src, 64);
            uint32_t run_n = ok ? wn_s : 0u, pmin = pidx;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t rid_o = (uint32_t)__shfl_down((int)rid, off, 64);      // (every lane takes part: a shuffle reads nothing from a lane that sits it out)
                const bool take = lane + off < 64 && rid_o == rid;                   // (runs are contiguous)
#pragma unroll
                for (int j = 0; j < 9; ++j) { const double o = __shfl_down(q[j], off, 64); if (take) q[j] += o; }
                const uint32_t on = (uint32_t)__shfl_down((int)run_n, off, 64), op = (uint32_t)__shfl_down((int)pmin, off, 64);
                if (take) { run_n += on; pmin = min(pmin, op); }
            }
            // The runs' first lanes hold the totals.  Their nine fp64 sums are not added lane by lane (nine instructions whose
            // lanes each hit a different node: every lane its own 64-byte request at the memory side) but TRANSPOSED through
            // LDS: consecutive lanes take consecutive sums of one node, so an instruction's 64 lanes cover seven nodes' 72
            // contiguous bytes each — a few requests instead of dozens (MI355X_MICROARCH, global float atomics: the memory
            // side is paid per 64-byte segment an instruction touches, not per lane).
            const bool lead = ok && head;
            const unsigned long long lm = __ballot(lead);
            const uint32_t n_lead = (uint32_t)__popcll(lm);
            const uint32_t li = (uint32_t)__popcll(lm & ((1ull << lane) - 1ull));     // this lane's place among the leading lanes
            double* const wq = &s_q[threadIdx.x >> 6][0];
            uint32_t* const wslot = &s_slot[threadIdx.x >> 6][0];
            if (lead) {
                bool inserted;
                uint32_t slot = find_or_insert(keys, cap_mask, key, inserted);
                if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); inserted = false; }
                wslot[li] = slot <= cap_mask ? slot : 0xFFFFFFFFu;
                if (slot <= cap_mask) {
                    app_slot = slot; app_new = inserted;
                    app_touch = mark && atomicExch(&touch_epoch[slot], epoch) != epoch;
#pragma unroll
                    for (int j = 0; j < 9; ++j) wq[li * 9 + j] = q[j];
                    NodeAcc* a = acc + slot;
                    atomicAdd(&a->count, run_n);
                    atomicMin(&a->first, pmin);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (the wave's LDS writes before its LDS reads below)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t e = (uint32_t)lane; e < n_lead * 9u; e += 64u) {
                const uint32_t hi = (e * 7282u) >> 16;                  // e / 9 for e < 576
                const uint32_t sl = wslot[hi];
                if (sl != 0xFFFFFFFFu) unsafeAtomicAdd(&acc[sl].s[e - hi * 9u], wq[e]);
            }
            __builtin_amdgcn_wave_barrier();                             // (the next iteration writes the same LDS)
        }
        // ---- the two lists (new nodes, touched nodes): ONE counter atomic per BLOCK and list.  A counter is one word, and
        // same-address atomics retire at ~90 per microsecond: one per wave — 2 048 for a 131 k-point frame — kept this kernel
        // busy for 20 of its 60 us.
        {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const unsigned long long m_new = __ballot(app_new), m_touch = __ballot(app_touch);
            if (lane == 0) { s_app[0][wave] = (uint32_t)__popcll(m_new); s_app[1][wave] = (uint32_t)__popcll(m_touch); }
            __syncthreads();
            if (threadIdx.x < 2) {
                uint32_t total = 0;
                for (int w = 0; w < kBlock / 64; ++w) total += s_app[threadIdx.x][w];
                uint32_t b = 0;
                if (total) b = atomicAdd(threadIdx.x == 0 ? &cnt->num_nodes : &cnt->n_touched, total);
                s_app[threadIdx.x][kBlock / 64] = b;
            }
            __syncthreads();
            uint32_t b_new = s_app[0][kBlock / 64], b_touch = s_app[1][kBlock / 64];
            for (int w = 0; w < wave; ++w) { b_new += s_app[0][w]; b_touch += s_app[1][w]; }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (app_new) {
                const uint32_t idx = b_new + (uint32_t)__popcll(m_new & below);
                node_slot[idx] = app_slot;
                index_of_slot[app_slot] = idx;
            }
            if (app_touch) touched[b_touch + (uint32_t)__popcll(m_touch & below)] = app_slot;
            __syncthreads();                                             // (s_app is written again by the next iteration)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_remove: the inverse of k_accumulate (intent of del2DMap, include/map2D.h:826-915: points leave their nodes, a node
// whose last point leaves is deleted).  The statistics are additive, so a point is taken away by subtracting its
// contribution about the same node centre; `first` (the node's place in the order) stays.  A point whose node does not
// exist, or is already empty, is counted in err_remove and changes nothing.
// ---------------------------------------------------------------------------------------------
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kBlock) k_remove(const float* __restrict__ xyz, uint64_t n, GridParams P,
                                                   const uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                                   Counters* __restrict__ cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float* p = xyz + i * STRIDE_FLOATS;
        const PointKey k = point_key(p[0], p[1], p[2], P.ox, P.oy, P.oz, P.grid_len, P.z_len);
        uint32_t slot = cap_mask + 1;
        if (k.ok) slot = find_slot(keys, cap_mask, pack_key(k.sx, k.sy, k.sz));
        if (slot > cap_mask) { atomicAdd(&cnt->err_remove, 1u); continue; }
        NodeAcc* a = acc + slot;
        const uint32_t before = atomicSub(&a->count, 1u);
        if (before == 0u) { atomicAdd(&a->count, 1u); atomicAdd(&cnt->err_remove, 1u); continue; }   // nothing left to take
        if (before == 1u) atomicAdd(&cnt->n_dead, 1u);
        const double v0 = (double)p[0] - axis_centre(k.sx, P.ox, P.grid_len), v1 = (double)p[1] - axis_centre(k.sy, P.oy, P.grid_len),
                     v2 = (double)p[2] - axis_centre(k.sz, P.oz, P.z_len);
        const double q[9] = {v0, v1, v2, v0 * v0, v0 * v1, v0 * v2, v1 * v1, v1 * v2, v2 * v2};
#pragma unroll
        for (int j = 0; j < 9; ++j) unsafeAtomicAdd(&a->s[j], -q[j]);
    }
}

// after an incremental accumulate: advance the device-side stream position
static __global__ void k_advance_stream(Counters* c, uint32_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c->stream_pos += n;
}

// after an accumulate with a caller-given base: the stream position is at least `v`
static __global__ void k_raise_stream(Counters* c, uint32_t v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c->stream_pos = max(c->stream_pos, v);
}

// ---------------------------------------------------------------------------------------------
// statistics exchange helpers (table growth, multi-GPU): compact export and additive merge
// ---------------------------------------------------------------------------------------------
static __global__ void k_stats_export(const uint64_t* __restrict__ keys, const NodeAcc* __restrict__ acc,
                               const uint32_t* __restrict__ node_slot, const Counters* __restrict__ cnt,
                               uint64_t* __restrict__ okey, double* __restrict__ osums, uint32_t* __restrict__ ocount,
                               uint32_t* __restrict__ ofirst) {
    const uint32_t n = cnt->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = node_slot[i];
        okey[i] = keys[s];
        const NodeAcc a = acc[s];
        for (int k = 0; k < 9; ++k) osums[9 * (uint64_t)i + k] = a.s[k];
        ocount[i] = a.count;
        ofirst[i] = a.first;
    }
}

static __global__ void __launch_bounds__(kBlock) k_stats_merge(uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                                        uint32_t* __restrict__ node_slot, uint32_t* __restrict__ index_of_slot,
                                                        const uint64_t* __restrict__ ikey,
                                                        const double* __restrict__ isums, const uint32_t* __restrict__ icount,
                                                        const uint32_t* __restrict__ ifirst, uint64_t n,
                                                        Counters* __restrict__ cnt) {
    __shared__ uint32_t s_bound;
    if (threadIdx.x == 0) s_bound = 0;
    __syncthreads();
    uint32_t bound = 0;   // one past the largest first_idx merged: the stream position is at least that
    const uint64_t n_round = (n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool live = i < n && ikey[i] != kEmptyKey && icount[i] != 0;
        bool inserted = false;
        uint32_t slot = cap_mask + 1;
        if (live) {
            slot = find_or_insert(keys, cap_mask, ikey[i], inserted);
            if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); inserted = false; }
        }
        append_new_nodes(inserted, slot, node_slot, index_of_slot, cnt);
        if (live && slot <= cap_mask) {
            NodeAcc* a = acc + slot;
            for (int k = 0; k < 9; ++k) unsafeAtomicAdd(&a->s[k], isums[9 * i + k]);
            atomicAdd(&a->count, icount[i]);
            const uint32_t f = ifirst[i];
            atomicMin(&a->first, f);
            if (f != 0xFFFFFFFFu) bound = max(bound, f + 1u);
        }
    }
    if (bound) atomicMax(&s_bound, bound);
    __syncthreads();
    if (threadIdx.x == 0 && s_bound) atomicMax(&cnt->stream_pos, s_bound);
}

}  // namespace gndt
