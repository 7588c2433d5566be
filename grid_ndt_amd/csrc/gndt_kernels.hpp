// gndt_kernels.hpp — HIP kernels of the NDT grid build for gfx950 (MI355X).
//
// Data layout in HBM (all owned by the handle, see gndt_api.hip):
//   keys[cap]      u64   open-addressing node table, kEmptyKey = free; linear probing
//   acc[cap]       NodeAcc (80 B): additive cell-local statistics of the node in that slot
//   col_keys[cap]  u64   column table (key with the z field cleared)
//   col_first[cap] u32   first-seen point index of the column (min over its nodes)
//   node_slot[C]   u32   compact list of occupied slots (unordered)
//   aux[cap]       SlotAux: fp32 mean-z for the slope test + label flags
//   out.*          SoA result rows in reference order (gndt_cells)
//
// Pipeline (strategy ATOMIC):
//   k_accumulate   points -> key -> find-or-insert -> fp64 atomics            (receiver.cpp:41-93)
//   k_scan_nodes   occupied slots -> node list, column table, mean-z           (map2D.h:611-627, part)
//   k_label_nodes  per node: slope label from z+-1 neighbours, sort key        (map2D.h:66-108)
//   radix sort     nodes by (column first-seen, node first-seen)               (morton_list / multimap order)
//   k_emit_nodes   per rank: mean, scatter, eigen -> SoA rows                  (map2D.h:621-627, 110-133)
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
    uint32_t num_nodes;
    uint32_t num_columns;
    uint32_t num_slopes;
    uint32_t err_key_range;   // points outside the key range
    uint32_t err_table_full;  // inserts that found no slot
    uint32_t pad[3];
};

struct GridParams {
    float ox, oy, oz;
    float grid_len, z_len, slope_interval;
    int demand, min_points;
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
__device__ __forceinline__ uint32_t find_or_insert(uint64_t* __restrict__ keys, uint32_t cap_mask, uint64_t key) {
    uint32_t slot = (uint32_t)mix64(key) & cap_mask;
    for (uint32_t probe = 0; probe <= cap_mask; ++probe) {
        uint64_t k = keys[slot];
        if (k == key) return slot;
        if (k == kEmptyKey) {
            unsigned long long old = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                                               (unsigned long long)key);
            if (old == kEmptyKey || old == key) return slot;
        }
        slot = (slot + 1) & cap_mask;
    }
    return cap_mask + 1;
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
__global__ void k_clear_all(uint64_t* keys, NodeAcc* acc, uint64_t* col_keys, uint32_t* col_first, uint32_t cap) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        keys[i] = kEmptyKey;
        col_keys[i] = kEmptyKey;
        col_first[i] = 0xFFFFFFFFu;
        NodeAcc a;
        for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
        a.count = 0; a.first = 0xFFFFFFFFu;
        acc[i] = a;
    }
}

// k_clear_used: undo the previous build by visiting only its occupied slots (O(C), not O(cap)).
__global__ void k_clear_used(uint64_t* keys, NodeAcc* acc, uint64_t* col_keys, uint32_t* col_first,
                             const uint32_t* node_slot, const uint32_t* col_slot_of_node, const Counters* prev) {
    const uint32_t n = prev->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = node_slot[i];
        const uint32_t cs = col_slot_of_node[i];   // nodes of one column write the same values: benign
        keys[slot] = kEmptyKey;
        NodeAcc a;
        for (int k = 0; k < 9; ++k) a.s[k] = 0.0;
        a.count = 0; a.first = 0xFFFFFFFFu;
        acc[slot] = a;
        col_keys[cs] = kEmptyKey;
        col_first[cs] = 0xFFFFFFFFu;
    }
}

// zero the per-finalize counters (node/column/slope); `all` also clears the sticky error counters
__global__ void k_zero_counters(Counters* c, int all) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        c->num_nodes = 0; c->num_columns = 0; c->num_slopes = 0;
        if (all) { c->err_key_range = 0; c->err_table_full = 0; }
    }
}

// ---------------------------------------------------------------------------------------------
// k_accumulate (strategy ATOMIC): one point per thread-iteration
//   uniformDivision (src/receiver.cpp:41-93) + transMortonXYZ (include/map2D.h:950-976), with the
//   per-node point list replaced by additive statistics.
// HBM roofline view: reads 12 (or 16) B/point; the 11 atomics per point execute at the memory side.
// When every lane of a wave holds the same key (the (0,0,0) padding of the reference's own clouds,
// SURVEY §4) the wave reduces in registers and issues one set of atomics.
// ---------------------------------------------------------------------------------------------
template <int STRIDE_FLOATS>
__global__ void __launch_bounds__(kBlock) k_accumulate(const float* __restrict__ xyz, uint64_t n, uint32_t first_base,
                                                       GridParams P, uint64_t* __restrict__ keys,
                                                       NodeAcc* __restrict__ acc, uint32_t cap_mask,
                                                       Counters* __restrict__ cnt) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ull;   // keep waves converged for the wave-uniform test
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool live = i < n;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (live) {
            const float* p = xyz + i * STRIDE_FLOATS;
            px = p[0]; py = p[1]; pz = p[2];
        }
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
        uint32_t pidx = first_base + (uint32_t)i;

        // wave-uniform key?  (all 64 lanes live, ok and equal)
        const uint64_t key0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(key >> 32)) << 32) |
                              (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)key);
        const bool uniform = __all(ok && key == key0);
        if (uniform) {
#pragma unroll
            for (int j = 0; j < 9; ++j) q[j] = wave_sum(q[j]);
            uint32_t pmin = pidx;
            for (int off = 32; off > 0; off >>= 1) pmin = min(pmin, (uint32_t)__shfl_down((int)pmin, off, 64));
            if ((threadIdx.x & 63) == 0) {
                uint32_t slot = find_or_insert(keys, cap_mask, key);
                if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); }
                else {
                    NodeAcc* a = acc + slot;
#pragma unroll
                    for (int j = 0; j < 9; ++j) unsafeAtomicAdd(&a->s[j], q[j]);
                    atomicAdd(&a->count, 64u);
                    atomicMin(&a->first, pmin);
                }
            }
        } else if (ok) {
            uint32_t slot = find_or_insert(keys, cap_mask, key);
            if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); }
            else {
                NodeAcc* a = acc + slot;
#pragma unroll
                for (int j = 0; j < 9; ++j) unsafeAtomicAdd(&a->s[j], q[j]);
                atomicAdd(&a->count, 1u);
                atomicMin(&a->first, pidx);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_scan_nodes: visit every slot; for occupied ones append to the node list, register the column
// (first-seen = min over its nodes), and store the fp32 mean-z the slope test compares.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_scan_nodes(const uint64_t* __restrict__ keys, const NodeAcc* __restrict__ acc,
                                                       uint64_t* __restrict__ col_keys, uint32_t* __restrict__ col_first,
                                                       uint32_t* __restrict__ node_slot, uint32_t* __restrict__ col_slot_of_node,
                                                       SlotAux* __restrict__ aux, uint32_t cap_mask, GridParams P,
                                                       Counters* __restrict__ cnt) {
    const uint32_t cap = cap_mask + 1;
    const uint32_t cap_round = (cap + 63) & ~63u;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < cap_round; s += gridDim.x * blockDim.x) {
        uint64_t key = (s < cap) ? keys[s] : kEmptyKey;
        const bool occ = key != kEmptyKey;
        // wave-aggregated append
        const unsigned long long m = __ballot(occ);
        uint32_t base = 0;
        const int lane = threadIdx.x & 63;
        if (m != 0ull) {
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cnt->num_nodes, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(m), 64);
        }
        if (!occ) continue;
        const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        node_slot[idx] = s;
        const NodeAcc a = acc[s];
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        SlotAux x;
        x.flags = 0;
        x.mean_z = 0.f;
        if (a.count >= (uint32_t)P.min_points) {
            x.mean_z = node_mean_z(a.count, a.s[2], axis_centre(sz, P.oz, P.z_len));
            x.flags = 1u;  // GNDT_FLAG_HAS_STATS
        }
        aux[s] = x;
        // column registration
        const uint64_t ck = column_key(key);
        uint32_t cs = (uint32_t)mix64(ck) & cap_mask;
        bool placed = false;
        for (uint32_t probe = 0; probe <= cap_mask && !placed; ++probe) {
            uint64_t k = col_keys[cs];
            if (k == ck) placed = true;
            else if (k == kEmptyKey) {
                unsigned long long old = atomicCAS((unsigned long long*)&col_keys[cs], (unsigned long long)kEmptyKey,
                                                   (unsigned long long)ck);
                if (old == kEmptyKey) placed = true;
                else if (old == ck) placed = true;
            }
            if (!placed) cs = (cs + 1) & cap_mask;
        }
        col_slot_of_node[idx] = cs;
        atomicMin(&col_first[cs], a.first);
    }
}

// ---------------------------------------------------------------------------------------------
// k_label_nodes: OcNode::isSlope (include/map2D.h:66-108) without the visiting order:
//   a neighbour's centroid is "already computed" iff it was first seen earlier in the same column
//   and has >= min_points points; otherwise the reference reads its zero-initialised centroid.
// Also emits the 64-bit sort key (column first-seen, node first-seen).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_label_nodes(const uint64_t* __restrict__ keys, const NodeAcc* __restrict__ acc,
                                                        const uint32_t* __restrict__ col_first,
                                                        const uint32_t* __restrict__ node_slot,
                                                        const uint32_t* __restrict__ col_slot_of_node,
                                                        SlotAux* __restrict__ aux, uint64_t* __restrict__ sort_key,
                                                        uint32_t* __restrict__ sort_val, uint32_t cap_mask, GridParams P,
                                                        Counters* __restrict__ cnt) {
    const uint32_t n = cnt->num_nodes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t s = node_slot[i];
        const uint64_t key = keys[s];
        const uint32_t my_first = acc[s].first;
        const uint32_t my_count = acc[s].count;
        const uint32_t cfirst = col_first[col_slot_of_node[i]];
        sort_key[i] = ((uint64_t)cfirst << 32) | (uint64_t)my_first;
        sort_val[i] = s;
        if (cfirst == my_first) atomicAdd(&cnt->num_columns, 1u);   // this node opened its column (morton_list entry)
        uint32_t flags = aux[s].flags;
        if (my_count >= (uint32_t)P.min_points) {
            bool slope = true, down = false;
            if (P.demand == 0) {
                int sx, sy, sz;
                unpack_key(key, sx, sy, sz);
                const float cz = aux[s].mean_z;
                bool up = false;
                // node one level up
                uint32_t t = find_slot(keys, cap_mask, pack_key(sx, sy, level_above(sz)));
                if (t <= cap_mask) {
                    const bool visited = acc[t].first < my_first && acc[t].count >= (uint32_t)P.min_points;
                    const float oz = visited ? aux[t].mean_z : 0.f;
                    if (fabsf(oz - cz) > P.slope_interval) up = true;
                }
                t = find_slot(keys, cap_mask, pack_key(sx, sy, level_below(sz)));
                if (t <= cap_mask) {
                    const bool visited = acc[t].first < my_first && acc[t].count >= (uint32_t)P.min_points;
                    const float oz = visited ? aux[t].mean_z : 0.f;
                    if (fabsf(oz - cz) > P.slope_interval) down = true;
                }
                slope = !up;
            }
            if (slope) {
                flags |= 2u;  // GNDT_FLAG_SLOPE
                if (down) flags |= 4u;
                atomicAdd(&cnt->num_slopes, 1u);
            }
        }
        aux[s].flags = flags;
    }
}

// ---------------------------------------------------------------------------------------------
// k_emit_nodes: one thread per output row (rank in reference order): mean, scatter, eigen.
// Algorithmic bytes: 80 B statistics in, 76 B result out per node.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_emit_nodes(const uint64_t* __restrict__ keys, const NodeAcc* __restrict__ acc,
                                                       const SlotAux* __restrict__ aux, const uint32_t* __restrict__ order,
                                                       OutView out, GridParams P, const Counters* __restrict__ cnt) {
    const uint32_t n = cnt->num_nodes;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const uint32_t s = order[r];
        const uint64_t key = keys[s];
        const NodeAcc a = acc[s];
        int sx, sy, sz;
        unpack_key(key, sx, sy, sz);
        out.sx[r] = sx; out.sy[r] = sy; out.sz[r] = sz;
        out.count[r] = a.count;
        out.first_idx[r] = a.first;
        out.flags[r] = aux[s].flags;
        NodeResult res;
        for (int k = 0; k < 3; ++k) { res.mean[k] = 0.f; res.normal[k] = 0.f; }
        for (int k = 0; k < 6; ++k) res.cov[k] = 0.f;
        res.rough = 0.f;
        if (a.count >= (uint32_t)P.min_points) {
            const double c[3] = {axis_centre(sx, P.ox, P.grid_len), axis_centre(sy, P.oy, P.grid_len),
                                 axis_centre(sz, P.oz, P.z_len)};
            finalize_node(a.count, a.s, c, res);
        }
        for (int k = 0; k < 3; ++k) { out.mean[3 * r + k] = res.mean[k]; out.normal[3 * r + k] = res.normal[k]; }
        for (int k = 0; k < 6; ++k) out.cov[6 * r + k] = res.cov[k];
        out.rough[r] = res.rough;
    }
}

// ---------------------------------------------------------------------------------------------
// statistics exchange helpers (multi-GPU): compact export and additive merge
// ---------------------------------------------------------------------------------------------
__global__ void k_stats_export(const uint64_t* __restrict__ keys, const NodeAcc* __restrict__ acc,
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

__global__ void k_stats_merge(uint64_t* __restrict__ keys, NodeAcc* __restrict__ acc, uint32_t cap_mask,
                              const uint64_t* __restrict__ ikey, const double* __restrict__ isums,
                              const uint32_t* __restrict__ icount, const uint32_t* __restrict__ ifirst, uint64_t n,
                              Counters* __restrict__ cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ikey[i];
        if (key == kEmptyKey || icount[i] == 0) continue;
        uint32_t slot = find_or_insert(keys, cap_mask, key);
        if (slot > cap_mask) { atomicAdd(&cnt->err_table_full, 1u); continue; }
        NodeAcc* a = acc + slot;
        for (int k = 0; k < 9; ++k) unsafeAtomicAdd(&a->s[k], isums[9 * i + k]);
        atomicAdd(&a->count, icount[i]);
        atomicMin(&a->first, ifirst[i]);
    }
}

}  // namespace gndt
