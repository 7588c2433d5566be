// gndt_cost.hpp — the cost-map flood over a finished grid: the immediate consumer
// of the grid-build path, on the GPU-resident result rows.
//
// Reference behaviour reproduced (each routine cites the statement whose RESULT it returns):
//   TwoDmap::computeCost      include/map2D.h:1285-1397   FIFO label-correcting flood from the goal slope
//   CollisionCheck / 3D       include/map2D.h:351-411, 414-474
//   AccessibleNeighbors       include/map2D.h:530-588, countReachable :262-337, countLRFB :197-259
//   countAngle :477-482, TravelCost :523-526, Slope::countUp :147-177, RobotSphere include/robot.h:38-46
//
// Why a level-synchronous flood gives the reference's numbers.  The reference pops a FIFO queue, so slopes are
// processed breadth-first: layer k = the slopes first reached from layer k-1.  Neighbours always lie in
// adjacent columns, so the slope graph is bipartite (parity of the column coordinates) and two slopes of one
// layer are never neighbours.  A slope is expanded once, with the h it has when it is popped, i.e. after all of
// layer k-1 relaxed it and before anything of layer k+1 does; later improvements change its stored h but are
// not propagated (map2D.h:1331-1336: it is pushed only while in none of the three lists).  Therefore
//   h_pop(p)  = min over expanded q in layer k-1 with p accessible from q of  h_pop(q) + d(q,p)
//   h_final(p) = min(h_pop(p) or FLT_MAX if p collided, the same expression over layer k+1)
// and every one of these minima is over a SET, so one kernel launch per layer with an atomic min on the fp32
// bit pattern (h >= 0) reproduces h bit for bit, whatever the order of the threads.
//
// The per-slope logic is host-callable so that the CPU-only test tier runs the same code level by level
// (tests/host_math_shim.cpp); the kernels are at the bottom.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "gndt_math.hpp"

#if defined(__clang__)
#define GNDT_FP_STRICT _Pragma("clang fp contract(off)")
#else
#define GNDT_FP_STRICT   // g++ builds of the host shim pass -ffp-contract=off
#endif

namespace gndt {

struct Robot {   // include/robot.h:12, 38-46
    float r, reach, rough, angle;
};

constexpr int kCostMaxXY = 32767;        // mortonToXY decodes only up to here (Stopwatch.h:171-189)
constexpr uint32_t kNoColumn = 0xFFFFFFFFu;
constexpr int kRingCap = 256;            // slopes a serial checker's ring holds in the host shim (tests/host_math_shim.cpp)
constexpr int kRingCapMax = 1 << 17;     // the device grows a checker's scratch fourfold when a ring does not fit (a 1.3 m robot on
                                         //   0.1 m cells: 27 x 27 columns, every level of them with demand "true") up to this

struct CostView {
    // result rows in reference order (gndt_cells)
    const int32_t *sx, *sy, *sz;
    const float *mean, *normal, *rough;
    const uint32_t* flags;
    // row_ncol[r] = number of nodes of the column that STARTS at row r (its nodes are rows [r, r + row_ncol[r])), 0 elsewhere
    const uint32_t* row_ncol;
    // (sx, sy) -> first row of the column, open addressing
    const uint64_t* ctab_key;
    const uint32_t* ctab_val;
    uint32_t ctab_mask;
    // optional: nbr[4 * row + k] = first row of the row's k-th neighbour column (left, right, forward, back) or kNoColumn,
    // precomputed for every row so that the per-layer kernels do not probe the hash table (null: probe)
    const uint32_t* nbr;
    float slope_interval;
    int demand_true;
    // optional, like nbr: self[2 * row] = first row of the row's OWN column, self[2 * row + 1] = the next slope above the row in its
    // cell (map_slope is ascending in z) or kNoColumn — what every collision check asks first, found once per flood
    const uint32_t* self = nullptr;
};

GNDT_HD uint64_t column_pack(int sx, int sy) { return pack_key(sx, sy, 0); }

GNDT_HD uint32_t ctab_find(const CostView& V, int sx, int sy) {
    const uint64_t key = column_pack(sx, sy);
    uint32_t s = (uint32_t)mix64(key) & V.ctab_mask;
    for (uint32_t probe = 0; probe <= V.ctab_mask; ++probe) {
        const uint64_t k = V.ctab_key[s];
        if (k == key) return V.ctab_val[s];
        if (k == kEmptyKey) return kNoColumn;
        s = (s + 1) & V.ctab_mask;
    }
    return kNoColumn;
}

// one step along an axis in signed cell coordinates: there is no cell 0 (countLRFB's x==1 / y==1 cases,
// map2D.h:226-255, hand the step over to the mirrored quadrant with index 1)
GNDT_HD int step_skip0(int v, int d) {
    int r = v + d;
    if (r == 0) r += d;
    return r;
}

// map2D.h:477-482.  dot in fp32 as Eigen's fixed-size reduction sums it, norms through double pow/sqrt,
// the quotient rounded to fp32, acosf, degrees through a double division, folded at 90.
GNDT_HD float cost_angle(const float* n1, const float* n2) {
    GNDT_FP_STRICT
    const float p0 = n1[0] * n2[0], p1 = n1[1] * n2[1], p2 = n1[2] * n2[2];
    const float dot = p0 + (p1 + p2);
    const double a0 = (double)n1[0] * (double)n1[0], a1 = (double)n1[1] * (double)n1[1], a2 = (double)n1[2] * (double)n1[2];
    const double b0 = (double)n2[0] * (double)n2[0], b1 = (double)n2[1] * (double)n2[1], b2 = (double)n2[2] * (double)n2[2];
    const double l1 = sqrt((a0 + a1) + a2), l2 = sqrt((b0 + b1) + b2);
    const float res = (float)((double)dot / (l1 * l2));
    const float ac = acosf(res) * 180.0f;
    float an = (float)((double)ac / 3.14159265358979323846);
    if (an > 90.f) an = 180.f - an;
    return an;
}

// map2D.h:523-526
GNDT_HD float cost_travel(const float* cur, const float* des) {
    GNDT_FP_STRICT
    const float dx = cur[0] - des[0], dy = cur[1] - des[1], dz = cur[2] - des[2];
    const double x2 = (double)dx * (double)dx, y2 = (double)dy * (double)dy, z2 = (double)dz * (double)dz;
    return (float)sqrt((x2 + y2) + z2);
}

GNDT_HD bool row_has_slope(const CostView& V, uint32_t row) { return (V.flags[row] & 2u) != 0u; }

GNDT_HD uint32_t row_column(const CostView& V, uint32_t row) { return V.self ? V.self[2 * (size_t)row] : ctab_find(V, V.sx[row], V.sy[row]); }

// the next slope above `row` in its cell, or kNoColumn
GNDT_HD uint32_t row_next_above(const CostView& V, uint32_t row) {
    if (V.self) return V.self[2 * (size_t)row + 1];
    const uint32_t c = ctab_find(V, V.sx[row], V.sy[row]);
    if (c == kNoColumn) return kNoColumn;
    const uint32_t e = c + V.row_ncol[c];
    const int myz = V.sz[row];
    uint32_t next = kNoColumn;
    int next_z = 0;
    for (uint32_t t = c; t < e; ++t) {
        if (!row_has_slope(V, t) || V.sz[t] <= myz) continue;
        if (next == kNoColumn || V.sz[t] < next_z) { next = t; next_z = V.sz[t]; }
    }
    return next;
}

// Slope::countUp (map2D.h:147-177): a node one level up in the column whose centroid z differs by more than the
// interval; centroids of nodes without statistics are zero.  Only demand "true" evaluates it (lazily); with
// demand "slope" Slope::up is never assigned and stays false (map2D.h:636).
GNDT_HD bool row_up(const CostView& V, uint32_t row) {
    if (!V.demand_true) return false;
    const uint32_t c = row_column(V, row);
    if (c == kNoColumn) return false;
    const int zadd = level_above(V.sz[row]);
    const float mz = V.mean[3 * row + 2];
    const uint32_t b = c, e = c + V.row_ncol[c];
    for (uint32_t t = b; t < e; ++t) {
        if (V.sz[t] != zadd) continue;
        const float cz = (V.flags[t] & 1u) ? V.mean[3 * t + 2] : 0.f;
        if (fabsf(cz - mz) > V.slope_interval) return true;
    }
    return false;
}

// the three gates of countReachable (map2D.h:271-274): roughness, angle between normals, height difference
GNDT_HD bool cost_gates(const CostView& V, const Robot& R, uint32_t s, const float* normal, const float* mean) {
    if (!(V.rough[s] <= R.rough)) return false;
    if (!(cost_angle(V.normal + 3 * s, normal) <= R.angle)) return false;
    return fabsf(V.mean[3 * s + 2] - mean[2]) <= R.reach;
}

// The four neighbour columns in the reference's order: left, right, forward, back (map2D.h:540-546).
GNDT_HD void neighbour_columns(const CostView& V, uint32_t row, uint32_t col[4]) {
    if (V.nbr) {
        for (int k = 0; k < 4; ++k) col[k] = V.nbr[4 * (size_t)row + k];
        return;
    }
    const int sx = V.sx[row], sy = V.sy[row];
    col[0] = ctab_find(V, sx, step_skip0(sy, -1));
    col[1] = ctab_find(V, sx, step_skip0(sy, +1));
    col[2] = ctab_find(V, step_skip0(sx, +1), sy);
    col[3] = ctab_find(V, step_skip0(sx, -1), sy);
}

// CollisionCheck (map2D.h:351-411) and CollisionCheck3D (:414-474).  `ring` is scratch for `ring_cap` rows.
// Returns 1 = collide, 0 = free, -1 = the ring did not fit the scratch.
GNDT_HD int cost_collide(const CostView& V, const Robot& R, uint32_t slope, int ring_n, uint32_t* ring, int ring_cap = kRingCap) {
    GNDT_FP_STRICT
    if (row_up(V, slope)) return 1;
    const float mz = V.mean[3 * slope + 2];
    int n_all = 1, now_b = 0, now_e = 1;
    ring[0] = slope;
    for (int depth = 0; depth < ring_n; ++depth) {
        for (int i = now_b; i < now_e; ++i) {
            const uint32_t cur = ring[i];
            uint32_t col[4];
            neighbour_columns(V, cur, col);
            for (int k = 0; k < 4; ++k) {
                if (col[k] == kNoColumn) continue;
                const uint32_t b = col[k], e = b + V.row_ncol[b];
                for (uint32_t t = b; t < e; ++t) {
                    if (!row_has_slope(V, t)) continue;
                    // comand 3 (3D ring): every slope of the cell; comand 2.5: up == false and the three gates
                    if (!V.demand_true && !cost_gates(V, R, t, V.normal + 3 * cur, V.mean + 3 * cur)) continue;
                    bool seen = false;
                    for (int j = 0; j < n_all; ++j) seen = seen || (ring[j] == t);
                    if (seen) continue;
                    if (n_all >= ring_cap) return -1;
                    ring[n_all++] = t;
                }
            }
        }
        now_b = now_e;
        now_e = n_all;
    }
    for (int j = 0; j < n_all; ++j) {
        const float tz = V.mean[3 * ring[j] + 2];
        if (tz < mz && row_up(V, ring[j])) return 1;
        // map2D.h:388 / :451: `((a < b) + 2*r)` only asks for a non-zero number
        const float odd = (float)(tz < mz ? 1 : 0) + 2.f * R.r;
        if (tz > mz && odd != 0.f && (tz - mz > R.reach)) return 1;
    }
    // the next slope above in the same cell (map_slope is ascending in z)
    const uint32_t next = row_next_above(V, slope);
    if (next != kNoColumn) {
        const float nz = V.mean[3 * next + 2];
        if ((nz < mz + 2.f * R.r) && (nz - mz > R.reach)) return 1;
    }
    return 0;
}

// int n = (ceil(2*r/gridLen) - 1) / 2  in fp32, truncated (map2D.h:1310)
GNDT_HD int cost_ring_depth(float r, float grid_len) {
    GNDT_FP_STRICT
    const float c = ceilf(2.f * r / grid_len);
    return (int)((c - 1.f) / 2.f);
}

GNDT_HD uint32_t float_bits(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
GNDT_HD float bits_float(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }

// Expansion of one popped slope (map2D.h:1312-1345 / 1346-1378) towards ONE of its four neighbour cells
// (0 left, 1 right, 2 forward, 3 back): calls relax(neighbour_row, candidate_h) for every accessible slope of that
// cell and returns the number of checkList pushes.
template <typename Relax>
GNDT_HD uint32_t cost_expand_dir(const CostView& V, const Robot& R, uint32_t q, float hq, int dir, Relax relax) {
    GNDT_FP_STRICT
    uint32_t c;
    if (V.nbr) {
        c = V.nbr[4 * (size_t)q + dir];
    } else {
        const int sx = V.sx[q], sy = V.sy[q];
        c = dir == 0 ? ctab_find(V, sx, step_skip0(sy, -1))
          : dir == 1 ? ctab_find(V, sx, step_skip0(sy, +1))
          : dir == 2 ? ctab_find(V, step_skip0(sx, +1), sy)
                     : ctab_find(V, step_skip0(sx, -1), sy);
    }
    if (c == kNoColumn) return 0u;
    uint32_t checks = 0;
    const uint32_t b = c, e = c + V.row_ncol[c];
    for (uint32_t t = b; t < e; ++t) {
        if (!row_has_slope(V, t)) continue;
        ++checks;                                   // checkList.push_back (up is false / not consulted)
        if (!cost_gates(V, R, t, V.normal + 3 * q, V.mean + 3 * q)) continue;
        relax(t, hq + cost_travel(V.mean + 3 * q, V.mean + 3 * t));
    }
    return checks;
}

template <typename Relax>
GNDT_HD uint32_t cost_expand(const CostView& V, const Robot& R, uint32_t q, float hq, Relax relax) {
    uint32_t checks = 0;
    for (int dir = 0; dir < 4; ++dir) checks += cost_expand_dir(V, R, q, hq, dir, relax);
    return checks;
}

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct CostCounters {
    uint32_t frontier[3];     // sizes of the frontier of level L (index L % 3)
    uint32_t traversable, closed;
    uint32_t ring_overflow, range_error;
    int32_t goal_status;      // 0 flood started, 1 no cell at the goal, 2 no slope at the goal's level
    uint32_t levels;          // layers that held at least one slope
    uint32_t pad;
    unsigned long long check_pushes;
};

static __global__ void __launch_bounds__(256) k_cost_clear(uint32_t* __restrict__ h_bits, uint32_t* __restrict__ pushed,
                                                    uint32_t* __restrict__ state, uint32_t n, uint64_t* __restrict__ ctab_key,
                                                    uint32_t ctab_size, CostCounters* __restrict__ cc) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint32_t i = gid; i < n; i += gsz) { h_bits[i] = 0x7F7FFFFFu; pushed[i] = 0u; state[i] = 0u; }
    for (uint32_t i = gid; i < ctab_size; i += gsz) ctab_key[i] = kEmptyKey;
    if (gid == 0) {
        cc->frontier[0] = cc->frontier[1] = cc->frontier[2] = 0u;
        cc->traversable = cc->closed = cc->ring_overflow = cc->range_error = 0u;
        cc->goal_status = 1; cc->levels = 0u; cc->check_pushes = 0ull;
    }
}

// first row of every column -> hash table entry keyed by the column's (sx, sy)
static __global__ void __launch_bounds__(256) k_cost_columns(const int32_t* __restrict__ sx, const int32_t* __restrict__ sy,
                                                      const uint32_t* __restrict__ row_ncol, uint32_t num_rows,
                                                      uint64_t* __restrict__ ctab_key, uint32_t* __restrict__ ctab_val,
                                                      uint32_t ctab_mask, CostCounters* __restrict__ cc) {
    for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < num_rows; row += gridDim.x * blockDim.x) {
        if (row_ncol[row] == 0u) continue;              // not the first row of a column
        const int x = sx[row], y = sy[row];
        if (abs(x) > kCostMaxXY || abs(y) > kCostMaxXY) atomicAdd(&cc->range_error, 1u);
        const uint64_t key = column_pack(x, y);
        uint32_t s = (uint32_t)mix64(key) & ctab_mask;
        for (;;) {
            const unsigned long long old = atomicCAS((unsigned long long*)&ctab_key[s], (unsigned long long)kEmptyKey,
                                                     (unsigned long long)key);
            if (old == kEmptyKey) { ctab_val[s] = row; break; }
            s = (s + 1) & ctab_mask;
        }
    }
}

// the four neighbour columns of every row, its own column and the slope above it, once per flood (the layers then follow plain indices)
static __global__ void __launch_bounds__(256) k_cost_neighbours(CostView V, uint32_t num_rows, uint32_t* __restrict__ nbr, uint32_t* __restrict__ self) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < 4u * num_rows; t += gridDim.x * blockDim.x) {
        const uint32_t row = t >> 2, k = t & 3u;
        const int sx = V.sx[row], sy = V.sy[row];
        if (k == 0u) {                                  // (V.self is null here: both go through the hash table)
            self[2 * (size_t)row] = ctab_find(V, sx, sy);
            self[2 * (size_t)row + 1] = row_next_above(V, row);
        }
        nbr[t] = k == 0 ? ctab_find(V, sx, step_skip0(sy, -1))
               : k == 1 ? ctab_find(V, sx, step_skip0(sy, +1))
               : k == 2 ? ctab_find(V, step_skip0(sx, +1), sy)
                        : ctab_find(V, step_skip0(sx, -1), sy);
    }
}

// goal lookup (map2D.h:1294-1307): the slope of the goal's cell at the goal's level gets h = 0 and is queued
static __global__ void k_cost_goal(CostView V, int gx, int gy, int gz, uint32_t* __restrict__ h_bits, uint32_t* __restrict__ pushed,
                            uint32_t* __restrict__ frontier0, CostCounters* __restrict__ cc) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const uint32_t c = ctab_find(V, gx, gy);
    if (c == kNoColumn) { cc->goal_status = 1; return; }
    cc->goal_status = 2;
    const uint32_t b = c, e = c + V.row_ncol[c];
    for (uint32_t t = b; t < e; ++t) {
        if (V.sz[t] == gz && row_has_slope(V, t)) {
            h_bits[t] = 0u;
            pushed[t] = 1u;
            frontier0[0] = t;
            cc->frontier[0] = 1u;
            cc->goal_status = 0;
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CollisionCheck by a TEAM of lanes (round 4).  The ring of a slope is a SET: layer d + 1 = the slopes, not seen before, that pass
// the gates from SOME slope of layer d, and the verdict asks whether ANY member is too high (map2D.h:384-392) — nothing depends on
// the order the reference's list happens to hold them in.  So the (slope of the layer, neighbour cell) pairs of a ring layer are
// spread over the team's lanes, a member is claimed with one compare-and-swap in a hash set in LDS (which is also the "seen" test the
// serial walk spends a scan of the whole ring on), and the list itself is only appended to.  A serial checker's time is a chain of
// ~5 dependent memory round trips per neighbour cell, 20 cells for a ring of depth 2; the team pays the chain once per ring layer.
// (bridge_ground at its own parameters, 182 layers: 28.5 ms of flood with one lane per checker — 2.5 x the reference's own loop on
// one CPU core.)  All lanes of a team are lanes of ONE wavefront: they run in lockstep, and LDS serves a wavefront's requests in
// order, so what a lane wrote before team_sync() is what every lane reads after it.
// ---------------------------------------------------------------------------------------------
constexpr int kTeamRingCap = 1024;       // slopes a team's ring holds in LDS; a ring that does not fit is a whole wavefront's, in global scratch
constexpr int kTeamSetSize = 2048;       // (power of two, 2 x the ring)
constexpr uint32_t kSetEmpty = 0xFFFFFFFFu;

struct TeamCtl {
    uint32_t n;                          // ring members
    uint32_t flags;                      // 1 = collide, 2 = the ring does not fit
};

__device__ __forceinline__ void team_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Where a team keeps its ring (the list) and its set (open addressing over row numbers, kSetEmpty at rest: every check removes what
// it added).  In LDS: plain accesses, in order for the wavefront.
struct alignas(16) TeamLds {
    uint32_t ring[kTeamRingCap];
    alignas(16) uint32_t set[kTeamSetSize];
};
struct RingInLds {
    TeamLds* L;
    uint32_t ring_cap;                   // <= kTeamRingCap
    __device__ __forceinline__ uint32_t cap() const { return ring_cap; }
    __device__ __forceinline__ uint32_t mask() const { return (uint32_t)(kTeamSetSize - 1); }
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return L->ring[i]; }
    __device__ __forceinline__ void put(uint32_t i, uint32_t t) const { L->ring[i] = t; }
    __device__ __forceinline__ uint32_t cas(uint32_t s, uint32_t t) const { return atomicCAS(&L->set[s], kSetEmpty, t); }
    __device__ __forceinline__ uint32_t peek(uint32_t s) const { return __atomic_load_n(&L->set[s], __ATOMIC_RELAXED); }
    __device__ __forceinline__ void wipe(uint32_t s) const { L->set[s] = kSetEmpty; }
    __device__ __forceinline__ void sync() const { team_sync(); }
};
// In global memory: every access is a device-scope atomic (served by the L2, past the CU's cache, which a wavefront's own stores do
// not update), and a sync waits for the wavefront's stores to have arrived there before the lanes go on.
struct RingInGlobal {
    uint32_t *ring, *set;
    uint32_t ring_cap, set_mask;
    __device__ __forceinline__ uint32_t cap() const { return ring_cap; }
    __device__ __forceinline__ uint32_t mask() const { return set_mask; }
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return __hip_atomic_load(&ring[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void put(uint32_t i, uint32_t t) const { __hip_atomic_store(&ring[i], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ uint32_t cas(uint32_t s, uint32_t t) const { return atomicCAS(&set[s], kSetEmpty, t); }
    __device__ __forceinline__ uint32_t peek(uint32_t s) const { return __hip_atomic_load(&set[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void wipe(uint32_t s) const { __hip_atomic_store(&set[s], kSetEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void sync() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); team_sync(); }
};

__device__ __forceinline__ uint32_t team_set_home(uint32_t t, uint32_t mask) { return ((t * 0x9E3779B1u) >> 11) & mask; }

// true: t was not a member and is one now
template <typename Store>
__device__ __forceinline__ bool team_set_claim(const Store& S, uint32_t t) {
    uint32_t s = team_set_home(t, S.mask());
    for (;;) {
        const uint32_t old = S.cas(s, t);
        if (old == kSetEmpty) return true;
        if (old == t) return false;
        s = (s + 1u) & S.mask();
    }
}

// (every member is removed by exactly one lane, so a walk only ever passes slots that hold, or held, OTHER members)
template <typename Store>
__device__ __forceinline__ void team_set_release(const Store& S, uint32_t t) {
    uint32_t s = team_set_home(t, S.mask());
    while (S.peek(s) != t) s = (s + 1u) & S.mask();
    S.wipe(s);
}

// Same verdict as cost_collide(): 1 collide, 0 free, -1 the ring holds more than S.cap() slopes.  Called by all T lanes of a team
// (tl = 0 .. T-1) with the same arguments; returns the same value in all of them.  C is the team's control block in LDS.
template <int T, typename Store>
__device__ int cost_collide_team(const CostView& V, const Robot& R, uint32_t slope, int ring_n, const Store& S, TeamCtl& C, uint32_t tl) {
    GNDT_FP_STRICT
    if (row_up(V, slope)) return 1;
    const float mz = V.mean[3 * slope + 2];
    if (tl == 0u) { S.put(0u, slope); C.n = 1u; C.flags = 0u; (void)team_set_claim(S, slope); }
    S.sync();
    uint32_t now_b = 0u, now_e = 1u;
    for (int depth = 0; depth < ring_n; ++depth) {
        const uint32_t items = (now_e - now_b) * 4u;
        for (uint32_t w = tl; w < items; w += (uint32_t)T) {
            const uint32_t cur = S.get(now_b + (w >> 2)), k = w & 3u;
            uint32_t c;
            if (V.nbr) c = V.nbr[4 * (size_t)cur + k];
            else {
                const int sx = V.sx[cur], sy = V.sy[cur];
                c = k == 0u ? ctab_find(V, sx, step_skip0(sy, -1)) : k == 1u ? ctab_find(V, sx, step_skip0(sy, +1))
                  : k == 2u ? ctab_find(V, step_skip0(sx, +1), sy) : ctab_find(V, step_skip0(sx, -1), sy);
            }
            if (c == kNoColumn) continue;
            const uint32_t e = c + V.row_ncol[c];
            for (uint32_t t = c; t < e; ++t) {
                if (!row_has_slope(V, t)) continue;
                if (!V.demand_true && !cost_gates(V, R, t, V.normal + 3 * cur, V.mean + 3 * cur)) continue;
                if (__atomic_load_n(&C.flags, __ATOMIC_RELAXED) & 2u) break;        // (the set must not fill up either)
                if (!team_set_claim(S, t)) continue;
                const uint32_t pos = atomicAdd(&C.n, 1u);
                if (pos < S.cap()) S.put(pos, t);
                else atomicOr(&C.flags, 2u);
            }
        }
        S.sync();
        if (C.flags & 2u) break;
        now_b = now_e;
        now_e = C.n;
        if (now_b == now_e) break;                     // an empty ring layer: the deeper ones are empty, too
    }
    int res = 0;
    const bool overflow = (C.flags & 2u) != 0u;
    const uint32_t n_all = overflow ? 0u : C.n;
    if (overflow) res = -1;
    else {
        for (uint32_t j = tl; j < n_all; j += (uint32_t)T) {
            const uint32_t m = S.get(j);
            const float tz = V.mean[3 * m + 2];
            bool hit = tz < mz && row_up(V, m);
            // map2D.h:388 / :451: `((a < b) + 2*r)` only asks for a non-zero number
            const float odd = (float)(tz < mz ? 1 : 0) + 2.f * R.r;
            hit = hit || (tz > mz && odd != 0.f && (tz - mz > R.reach));
            if (hit) atomicOr(&C.flags, 1u);
        }
        team_sync();
        if (C.flags & 1u) res = 1;
        else {
            // the next slope above in the same cell (map_slope is ascending in z); the same loads in every lane of the team
            const uint32_t next = row_next_above(V, slope);
            if (next != kNoColumn) {
                const float nz = V.mean[3 * next + 2];
                if ((nz < mz + 2.f * R.r) && (nz - mz > R.reach)) res = 1;
            }
        }
    }
    // leave the set empty for the team's next slope
    if (overflow) { for (uint32_t j = tl; j <= S.mask(); j += (uint32_t)T) S.wipe(j); }
    else          { for (uint32_t j = tl; j < n_all; j += (uint32_t)T) team_set_release(S, S.get(j)); }
    S.sync();
    return res;
}

// one layer of the flood: collision check, then expansion, of every slope in the layer.  T lanes share a slope: 4 (one per neighbour
// cell of the expansion; the collision check without a ring is one lane's), 16 (rings that fit LDS: the check is the team's, the
// expansion its first four lanes') or 64 (rings that do not: ring and set in `scratch`, 3 * ring_cap words per wavefront, set part
// kSetEmpty at rest).  Layers are short, so the kernel is latency-bound and the serial work per lane counts.
template <int T>
static __global__ void __launch_bounds__(64) k_cost_level(CostView V, Robot R, int ring_n, uint32_t level, uint32_t* __restrict__ h_bits,
                                                   uint32_t* __restrict__ pushed, uint32_t* __restrict__ state,
                                                   const uint32_t* __restrict__ f_in, uint32_t* __restrict__ f_out,
                                                   uint32_t* scratch, uint32_t ring_cap, CostCounters* __restrict__ cc) {
    constexpr uint32_t kPerWave = 64u / (uint32_t)T;   // slopes a wave takes at a time
    __shared__ uint32_t s_one[T == 4 ? 16 : 1];        // a checker without a ring still lists the slope itself
    __shared__ TeamLds s_team[T == 16 ? kPerWave : 1];
    __shared__ TeamCtl s_ctl[kPerWave];
    const uint32_t n_in = cc->frontier[level % 3u];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cc->frontier[(level + 2u) % 3u] = 0u;          // the counter two layers ahead (was the previous layer's input)
        if (n_in) cc->levels = level + 1u;
    }
    if (blockIdx.x * kPerWave >= n_in) return;         // (also: n_in == 0)
    uint32_t* out_count = &cc->frontier[(level + 1u) % 3u];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t dir = tid % (uint32_t)T, teams = (gridDim.x * blockDim.x) / (uint32_t)T;
    const int lane = (int)(threadIdx.x & 63u), leader = lane & ~(T - 1);
    if constexpr (T == 16) {
        for (uint32_t j = threadIdx.x; j < kPerWave * (uint32_t)(kTeamSetSize / 4); j += blockDim.x)
            reinterpret_cast<uint4*>(s_team[j / (uint32_t)(kTeamSetSize / 4)].set)[j % (uint32_t)(kTeamSetSize / 4)] =
                make_uint4(kSetEmpty, kSetEmpty, kSetEmpty, kSetEmpty);
        __syncthreads();
    }
    uint32_t trav = 0, closed = 0, checks = 0;
    // The loop is wave-uniform (a wave's teams take consecutive slopes of the layer, lanes past the end sit idle), so that the
    // slopes a wave pushes are appended with ONE atomic on the layer's counter: the counter is one word, same-address atomics
    // retire at ~90 per microsecond at the memory side, and a layer of a few hundred slopes used to add one per pushed slope
    // and three more per expanded slope (the statistics below) — most of a layer's 13-18 us.
    for (uint32_t i0 = (tid >> 6) * kPerWave; i0 < n_in; i0 += teams) {
        const uint32_t i = i0 + (uint32_t)lane / (uint32_t)T;
        const bool live = i < n_in;
        const uint32_t q = live ? f_in[i] : 0u;
        int hit = 0;
        if constexpr (T == 16) {
            if (live) hit = cost_collide_team<T>(V, R, q, ring_n, RingInLds{&s_team[(uint32_t)lane / (uint32_t)T], ring_cap}, s_ctl[(uint32_t)lane / (uint32_t)T], dir);
        } else if constexpr (T == 64) {
            uint32_t* mine = scratch + (size_t)blockIdx.x * 3u * (size_t)ring_cap;
            if (live) hit = cost_collide_team<T>(V, R, q, ring_n, RingInGlobal{mine, mine + ring_cap, ring_cap, 2u * ring_cap - 1u}, s_ctl[0], dir);
        } else {
            if (live && dir == 0u) hit = cost_collide(V, R, q, 0, &s_one[threadIdx.x >> 2], 1);
            hit = __shfl(hit, leader, 64);
        }
        constexpr uint32_t kKeep = 4;                  // pushes a lane keeps for the wave's append (more go out one by one)
        uint32_t mine[kKeep] = {0u, 0u, 0u, 0u}, np = 0;
        if (live && dir < 4u) {
            if (hit < 0) { if (dir == 0u) atomicAdd(&cc->ring_overflow, 1u); }
            else if (hit) {
                if (dir == 0u) {
                    h_bits[q] = 0x7F7FFFFFu;          // Q.front()->h = FLT_MAX (map2D.h:1340)
                    state[q] = 2u;
                    ++closed;
                }
            } else {
                if (dir == 0u) { state[q] = 1u; ++trav; }
                const float hq = bits_float(h_bits[q]);
                checks += cost_expand_dir(V, R, q, hq, (int)dir, [&](uint32_t t, float cand) {
                    const uint32_t cb = float_bits(cand);
                    const uint32_t old = atomicMin(&h_bits[t], cb);
                    if (old > cb && atomicCAS(&pushed[t], 0u, 1u) == 0u) {
                        if (np == 0u) mine[0] = t; else if (np == 1u) mine[1] = t; else if (np == 2u) mine[2] = t; else if (np == 3u) mine[3] = t;
                        else f_out[atomicAdd(out_count, 1u)] = t;
                        ++np;
                    }
                });
            }
        }
        const uint32_t kept = min(np, kKeep);
        uint32_t incl = kept;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        if (total) {                                   // (wave-uniform)
            uint32_t base = 0;
            if (lane == 63) base = atomicAdd(out_count, total);
            base = (uint32_t)__shfl((int)base, 63, 64) + incl - kept;
            if (kept > 0u) f_out[base] = mine[0];
            if (kept > 1u) f_out[base + 1u] = mine[1];
            if (kept > 2u) f_out[base + 2u] = mine[2];
            if (kept > 3u) f_out[base + 3u] = mine[3];
        }
    }
    // the layer's statistics: one atomic per wave and counter
    for (int o = 32; o > 0; o >>= 1) {
        trav += (uint32_t)__shfl_down((int)trav, o, 64); closed += (uint32_t)__shfl_down((int)closed, o, 64);
        checks += (uint32_t)__shfl_down((int)checks, o, 64);
    }
    if (lane == 0) {
        if (trav) atomicAdd(&cc->traversable, trav);
        if (closed) atomicAdd(&cc->closed, closed);
        if (checks) atomicAdd(&cc->check_pushes, (unsigned long long)checks);
    }
}
#endif  // __HIPCC__

}  // namespace gndt
