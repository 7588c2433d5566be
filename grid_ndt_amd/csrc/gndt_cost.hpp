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
constexpr int kRingCap = 256;            // slopes a collision ring may hold at first (per checker scratch); the host doubles it when a
constexpr int kRingCapMax = 1 << 15;     //   ring does not fit (a 1.3 m robot on 0.1 m cells: 27 x 27 columns) up to this

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

// Slope::countUp (map2D.h:147-177): a node one level up in the column whose centroid z differs by more than the
// interval; centroids of nodes without statistics are zero.  Only demand "true" evaluates it (lazily); with
// demand "slope" Slope::up is never assigned and stays false (map2D.h:636).
GNDT_HD bool row_up(const CostView& V, uint32_t row) {
    if (!V.demand_true) return false;
    const uint32_t c = ctab_find(V, V.sx[row], V.sy[row]);
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
    const uint32_t c = ctab_find(V, V.sx[slope], V.sy[slope]);
    if (c == kNoColumn) return 0;
    const uint32_t b = c, e = c + V.row_ncol[c];
    const int myz = V.sz[slope];
    uint32_t next = kNoColumn;
    int next_z = 0;
    for (uint32_t t = b; t < e; ++t) {
        if (!row_has_slope(V, t) || V.sz[t] <= myz) continue;
        if (next == kNoColumn || V.sz[t] < next_z) { next = t; next_z = V.sz[t]; }
    }
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

// the four neighbour columns of every row, once per flood (the layers then follow plain indices)
static __global__ void __launch_bounds__(256) k_cost_neighbours(CostView V, uint32_t num_rows, uint32_t* __restrict__ nbr) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < 4u * num_rows; t += gridDim.x * blockDim.x) {
        const uint32_t row = t >> 2, k = t & 3u;
        const int sx = V.sx[row], sy = V.sy[row];
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

// one layer of the flood: collision check, then expansion, of every slope in the layer.  Four lanes share a slope
// (one per neighbour cell): layers are short, so the kernel is latency-bound and the serial work per lane counts.
static __global__ void __launch_bounds__(64) k_cost_level(CostView V, Robot R, int ring_n, uint32_t level, uint32_t* __restrict__ h_bits,
                                                   uint32_t* __restrict__ pushed, uint32_t* __restrict__ state,
                                                   const uint32_t* __restrict__ f_in, uint32_t* __restrict__ f_out,
                                                   uint32_t* __restrict__ ring_scratch, int ring_cap, CostCounters* __restrict__ cc) {
    const uint32_t n_in = cc->frontier[level % 3u];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cc->frontier[(level + 2u) % 3u] = 0u;          // the counter two layers ahead (was the previous layer's input)
        if (n_in) cc->levels = level + 1u;
    }
    if (n_in == 0u) return;
    uint32_t* out_count = &cc->frontier[(level + 1u) % 3u];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t quad = tid >> 2, dir = tid & 3u, quads = (gridDim.x * blockDim.x) >> 2;
    const int lane = (int)(threadIdx.x & 63u), leader = lane & ~3;
    uint32_t trav = 0, closed = 0, checks = 0;
    // The loop is wave-uniform (a wave's 16 quads take 16 consecutive slopes of the layer, lanes past the end sit idle), so that
    // the slopes a wave pushes are appended with ONE atomic on the layer's counter: the counter is one word, same-address atomics
    // retire at ~90 per microsecond at the memory side, and a layer of a few hundred slopes used to add one per pushed slope
    // and three more per expanded slope (the statistics below) — most of a layer's 13-18 us.
    for (uint32_t i0 = (tid >> 6) * 16u; i0 < n_in; i0 += quads) {
        const uint32_t i = i0 + ((uint32_t)lane >> 2);
        const bool live = i < n_in;
        const uint32_t q = live ? f_in[i] : 0u;
        int hit = 0;
        if (live && dir == 0u) hit = cost_collide(V, R, q, ring_n, ring_scratch + (size_t)quad * (size_t)ring_cap, ring_cap);
        hit = __shfl(hit, leader, 64);
        constexpr uint32_t kKeep = 4;                  // pushes a lane keeps for the wave's append (more go out one by one)
        uint32_t mine[kKeep] = {0u, 0u, 0u, 0u}, np = 0;
        if (live) {
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
