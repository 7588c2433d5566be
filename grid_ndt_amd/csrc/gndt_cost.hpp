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
// and every one of these minima is over a SET, so a layer worked on by any number of threads in any order, with an
// atomic min on the fp32 bit pattern (h >= 0), reproduces h bit for bit — one launch per layer, or one workgroup
// walking layer after layer with a barrier between them (k_cost_level / k_cost_flood_wg at the bottom).  CollisionCheck's
// verdict is found for every slope before the flood ("CollisionCheck without walking rings" below).
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
constexpr int kRingCap = 256;            // slopes the ring of the serial walk holds in the host shim (tests/host_math_shim.cpp)

// What a popped slope does to ONE of its four neighbour cells, found for every slope BEFORE the flood (round 5): neither the gates
// (roughness, angle, height: map2D.h:271-274) nor TravelCost (:523-526) depend on h, so the layer-by-layer walk — short layers, one
// dependent round trip after the other — only adds and takes minima.  Up to two accessible slopes of the cell are named here with
// their travel cost (a cell rarely holds more within the robot's reach); kEdgeMore sends the flood through cost_expand_column.
struct CostEdge {
    uint32_t c;        // first row of the neighbour column (kNoColumn: none)
    uint32_t info;     // bits 0-7 / 8-15: the accessible rows, as offsets from c; 16-17: how many of them (0..2); 18: kEdgeMore;
                       //   20-31: the cell's checkList pushes (map2D.h:1320-1323: its slopes)
    float d0, d1;      // TravelCost to them
};
constexpr uint32_t kEdgeMore = 1u << 18;

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
    // optional: nbr[8 * row + 2 * k] = first row of the row's k-th neighbour column (left, right, forward, back) or kNoColumn,
    // nbr[8 * row + 2 * k + 1] = the nodes of that column (one 8-byte load names the rows to look at), precomputed for every row so
    // that the per-layer kernels neither probe the hash table nor read row_ncol (null: probe)
    const uint32_t* nbr;
    float slope_interval;
    int demand_true;
    // optional, like nbr: self[2 * row] = first row of the row's OWN column, self[2 * row + 1] = 1 if the next slope above the row in
    // its cell (map_slope is ascending in z) is in the robot's way (map2D.h:394-410) — what every collision check asks, answered once
    // per flood (for that flood's robot)
    const uint32_t* self = nullptr;
    // optional: edges[4 * row + k] of every row that holds a slope (see CostEdge)
    const CostEdge* edges = nullptr;
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

// Is the next slope above `row` in its cell in the robot's way?  (CollisionCheck's last test, map2D.h:394-410 / :457-473.)
// c: the first row of the row's own column (kNoColumn: none).
GNDT_HD bool row_above_hits_in(const CostView& V, const Robot& R, uint32_t row, uint32_t c) {
    GNDT_FP_STRICT
    if (c == kNoColumn) return false;
    const uint32_t e = c + V.row_ncol[c];
    const int myz = V.sz[row];
    uint32_t next = kNoColumn;
    int next_z = 0;
    for (uint32_t t = c; t < e; ++t) {
        if (!((V.flags[t] & 2u) != 0u) || V.sz[t] <= myz) continue;
        if (next == kNoColumn || V.sz[t] < next_z) { next = t; next_z = V.sz[t]; }
    }
    if (next == kNoColumn) return false;
    const float mz = V.mean[3 * row + 2], nz = V.mean[3 * next + 2];
    return (nz < mz + 2.f * R.r) && (nz - mz > R.reach);
}
GNDT_HD bool row_above_hits(const CostView& V, const Robot& R, uint32_t row) {
    if (V.self) return V.self[2 * (size_t)row + 1] != 0u;
    return row_above_hits_in(V, R, row, ctab_find(V, V.sx[row], V.sy[row]));
}

// Slope::countUp (map2D.h:147-177): a node one level up in the column whose centroid z differs by more than the
// interval; centroids of nodes without statistics are zero.  Only demand "true" evaluates it (lazily); with
// demand "slope" Slope::up is never assigned and stays false (map2D.h:636).
GNDT_HD bool row_up_in(const CostView& V, uint32_t row, uint32_t c) {
    if (!V.demand_true) return false;
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
GNDT_HD bool row_up(const CostView& V, uint32_t row) {
    if (!V.demand_true) return false;
    return row_up_in(V, row, row_column(V, row));
}

// the three gates of countReachable (map2D.h:271-274): roughness, angle between normals, height difference
GNDT_HD bool cost_gates(const CostView& V, const Robot& R, uint32_t s, const float* normal, const float* mean) {
    if (!(V.rough[s] <= R.rough)) return false;
    if (!(cost_angle(V.normal + 3 * s, normal) <= R.angle)) return false;
    return fabsf(V.mean[3 * s + 2] - mean[2]) <= R.reach;
}

// The k-th neighbour column in the reference's order — left, right, forward, back (map2D.h:540-546): its first row and node count.
GNDT_HD void neighbour_column(const CostView& V, uint32_t row, uint32_t k, uint32_t& c, uint32_t& ncol) {
    if (V.nbr) {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint2 e = reinterpret_cast<const uint2*>(V.nbr)[4 * (size_t)row + k];
        c = e.x; ncol = e.y;
#else
        c = V.nbr[8 * (size_t)row + 2 * k]; ncol = V.nbr[8 * (size_t)row + 2 * k + 1];
#endif
        return;
    }
    const int sx = V.sx[row], sy = V.sy[row];
    c = k == 0u ? ctab_find(V, sx, step_skip0(sy, -1)) : k == 1u ? ctab_find(V, sx, step_skip0(sy, +1))
      : k == 2u ? ctab_find(V, step_skip0(sx, +1), sy) : ctab_find(V, step_skip0(sx, -1), sy);
    ncol = c == kNoColumn ? 0u : V.row_ncol[c];
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
            for (uint32_t k = 0; k < 4u; ++k) {
                uint32_t b, ncol;
                neighbour_column(V, cur, k, b, ncol);
                if (b == kNoColumn) continue;
                const uint32_t e = b + ncol;
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
    return row_above_hits(V, R, slope) ? 1 : 0;
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
// (nq, mq: normal and centroid of q; c, ncol: its neighbour column in this direction — read by the caller, on the device together
//  with everything else that only needs q)
template <typename Relax>
GNDT_HD uint32_t cost_expand_column(const CostView& V, const Robot& R, float hq, const float* nq, const float* mq, uint32_t c, uint32_t ncol, Relax relax) {
    GNDT_FP_STRICT
    if (c == kNoColumn) return 0u;
    uint32_t checks = 0;
    for (uint32_t t = c; t < c + ncol; ++t) {
        // everything the gates may ask of the row, requested at once (one round trip instead of one per gate; here — a slope's few
        // neighbour rows, most of which pass — that is a gain, 7.8 against 8.4 us per layer; in the rings' loop it is a loss)
        const uint32_t fl = V.flags[t];
        const float rg = V.rough[t];
        const float tn[3] = {V.normal[3 * t], V.normal[3 * t + 1], V.normal[3 * t + 2]};
        const float tm[3] = {V.mean[3 * t], V.mean[3 * t + 1], V.mean[3 * t + 2]};
        if (!(fl & 2u)) continue;
        ++checks;                                   // checkList.push_back (up is false / not consulted)
        // the three gates of countReachable (map2D.h:271-274): roughness, angle between normals, height difference
        if (!(rg <= R.rough)) continue;
        if (!(cost_angle(tn, nq) <= R.angle)) continue;
        if (!(fabsf(tm[2] - mq[2]) <= R.reach)) continue;
        relax(t, hq + cost_travel(mq, tm));
    }
    return checks;
}
template <typename Relax>
GNDT_HD uint32_t cost_expand_dir(const CostView& V, const Robot& R, uint32_t q, float hq, int dir, Relax relax) {
    uint32_t c, ncol;
    neighbour_column(V, q, (uint32_t)dir, c, ncol);
    return cost_expand_column(V, R, hq, V.normal + 3 * q, V.mean + 3 * q, c, ncol, relax);
}

template <typename Relax>
GNDT_HD uint32_t cost_expand(const CostView& V, const Robot& R, uint32_t q, float hq, Relax relax) {
    uint32_t checks = 0;
    for (int dir = 0; dir < 4; ++dir) checks += cost_expand_dir(V, R, q, hq, dir, relax);
    return checks;
}

// The record of slope q's k-th neighbour cell (c, ncol: neighbour_column): the same loop as cost_expand_column, with the relaxation
// left for the flood.
// STEPS (floods with collision rings): also the ring's steps of q into the cell (ring_step_mask below: bit j = row c + j for the first
// kStepRows rows) from the same walk — the ring asks the same three gates (comand 2.5) or takes every slope (comand 3).
struct CostEdgeAndSteps { CostEdge e; uint32_t steps; };
template <bool STEPS>
GNDT_HD CostEdgeAndSteps cost_edge_walk(const CostView& V, const Robot& R, uint32_t q, uint32_t c, uint32_t ncol) {
    GNDT_FP_STRICT
    CostEdge e{c, 0u, 0.f, 0.f};
    uint32_t steps = 0u;
    if (c == kNoColumn) return CostEdgeAndSteps{e, steps};
    const float* nq = V.normal + 3 * (size_t)q;
    const float* mq = V.mean + 3 * (size_t)q;
    uint32_t checks = 0, np = 0, j0 = 0, j1 = 0;
    bool more = false;
    for (uint32_t j = 0; j < ncol; ++j) {
        const uint32_t t = c + j;
        if (!(V.flags[t] & 2u)) continue;
        ++checks;
        if (STEPS && V.demand_true && j < 32u) steps |= 1u << j;
        if (!(V.rough[t] <= R.rough)) continue;
        if (!(cost_angle(V.normal + 3 * (size_t)t, nq) <= R.angle)) continue;
        if (!(fabsf(V.mean[3 * (size_t)t + 2] - mq[2]) <= R.reach)) continue;
        if (STEPS && !V.demand_true && j < 32u) steps |= 1u << j;
        const float d = cost_travel(mq, V.mean + 3 * (size_t)t);
        if (np == 0u && j < 256u) { j0 = j; e.d0 = d; }
        else if (np == 1u && j < 256u) { j1 = j; e.d1 = d; }
        else more = true;
        ++np;
    }
    if (checks > 4095u) more = true;
    e.info = more ? kEdgeMore : (j0 | (j1 << 8) | (np << 16) | (checks << 20));
    return CostEdgeAndSteps{e, steps};
}
GNDT_HD CostEdge cost_edge_record(const CostView& V, const Robot& R, uint32_t q, uint32_t c, uint32_t ncol) {
    return cost_edge_walk<false>(V, R, q, c, ncol).e;
}

// Expansion towards one neighbour cell from its record (the record must not carry kEdgeMore): relax(row, candidate h) as
// cost_expand_column calls it, the cell's checkList pushes returned.
template <typename Relax>
GNDT_HD uint32_t cost_expand_record(const CostEdge& e, float hq, Relax relax) {
    GNDT_FP_STRICT
    const uint32_t n = (e.info >> 16) & 3u;
    if (n > 0u) relax(e.c + (e.info & 0xFFu), hq + e.d0);
    if (n > 1u) relax(e.c + ((e.info >> 8) & 0xFFu), hq + e.d1);
    return e.info >> 20;
}

// ---------------------------------------------------------------------------------------------
// CollisionCheck without walking rings (round 4).  The ring of a slope q (map2D.h:351-411 / 414-474) is the set R_n(q) of slopes
// within n steps of q, a step going from a slope to a slope of one of its four neighbour cells that passes the gates seen from it
// (comand 2.5) or to any slope there (comand 3), and the verdict asks whether SOME member lies more than the robot's reach above q
// — or, comand 3, has `up` set and lies below q.  Both are questions about an extreme over the set, and the set is a union:
//     R_n(q) = {q}  +  the R_(n-1)(p) of q's steps p          =>          max z over R_n(q) = max(z(q), max over p of [max z over R_(n-1)(p)])
// so n rounds of "take the extreme over your steps" over ALL slopes of the map at once — the steps of a slope a bit mask per
// neighbour cell, computed once — give every slope's answer, exactly (the verdict's fp32 subtraction is monotone in the member's z,
// so the highest member decides).  The flood then reads one word per slope.  A ring is never listed: no capacity, no retry, and
// the cost does not grow with the ring's size (27 x 27 cells of a 1.3 m robot on 0.1 m cells: 13 rounds).
// The walk itself is kept above (cost_collide) as the statement of what is computed; the CPU tier checks these rounds against it
// for every slope of random maps (tests/test_cost_map.py), and runs it against the oracle.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kStepRows = 32;       // rows of a neighbour cell that fit its step mask; taller cells: the gates again in every round

// may the ring step from slope `row` to row t of a neighbour cell?
GNDT_HD bool ring_step_ok(const CostView& V, const Robot& R, uint32_t row, uint32_t t) {
    if (!(V.flags[t] & 2u)) return false;
    if (V.demand_true) return true;                    // comand 3 (3D ring): every slope of the cell
    return cost_gates(V, R, t, V.normal + 3 * row, V.mean + 3 * row);
}

// the steps of slope `row` into its k-th neighbour cell (first row c, ncol rows): bit j = row c + j, for the first kStepRows rows
GNDT_HD uint32_t ring_step_mask(const CostView& V, const Robot& R, uint32_t row, uint32_t c, uint32_t ncol) {
    uint32_t m = 0u;
    if (c == kNoColumn) return m;
    for (uint32_t j = 0; j < ncol && j < kStepRows; ++j) m |= ring_step_ok(V, R, row, c + j) ? 1u << j : 0u;
    return m;
}

// round 0 of a slope's extremes: the slope itself.  hi: the highest centroid in the ring; lo: the lowest of the members whose `up`
// is set (comand 3 consults it: map2D.h:447), FLT_MAX while there is none
GNDT_HD void ring_round0(const CostView& V, uint32_t row, bool up, float& hi, float& lo) {
    hi = V.mean[3 * row + 2];
    lo = up ? V.mean[3 * row + 2] : FLT_MAX;
}

// one round, one neighbour cell: the extremes over the steps of `row` into it
GNDT_HD void ring_round_cell(const CostView& V, const Robot& R, uint32_t row, uint32_t c, uint32_t ncol, uint32_t mask,
                             const float* hi_in, const float* lo_in, float& hi, float& lo) {
    if (c == kNoColumn) return;
    while (mask) {
        uint32_t j = 0;
        while (!((mask >> j) & 1u)) ++j;
        mask &= mask - 1u;
        hi = fmaxf(hi, hi_in[c + j]);
        if (V.demand_true) lo = fminf(lo, lo_in[c + j]);
    }
    for (uint32_t j = kStepRows; j < ncol; ++j) {      // (a cell of more than kStepRows nodes)
        if (!ring_step_ok(V, R, row, c + j)) continue;
        hi = fmaxf(hi, hi_in[c + j]);
        if (V.demand_true) lo = fminf(lo, lo_in[c + j]);
    }
}

// map2D.h:384-392 / 447-455 over the whole ring at once: a member with `up` below the slope (comand 3; lo is FLT_MAX otherwise), or a
// member more than the reach above it (`(a < b) + 2 r` only asks for a non-zero number, and r > 0 where there is a ring)
GNDT_HD bool ring_verdict(const CostView& V, const Robot& R, uint32_t row, float hi, float lo) {
    GNDT_FP_STRICT
    const float mz = V.mean[3 * row + 2];
    return (V.demand_true && lo < mz) || (hi > mz && (hi - mz > R.reach));
}

// what of CollisionCheck does not need the ring: Slope::up of the slope itself (comand 3, map2D.h:417) and the next slope above it
// in its cell (:394-410)
GNDT_HD bool ring_free_verdict(const CostView& V, const Robot& R, uint32_t row, bool up) { return up || row_above_hits(V, R, row); }

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct CostCounters {
    uint32_t frontier[3];     // sizes of the frontier of level L (index L % 3)
    uint32_t traversable, closed;
    uint32_t pad0, range_error;
    int32_t goal_status;      // 0 flood started, 1 no cell at the goal, 2 no slope at the goal's level
    uint32_t levels;          // layers that held at least one slope
    uint32_t pad;
    unsigned long long check_pushes;
    uint32_t wg_layers;       // layers walked by the one-workgroup kernel so far: a kernel's layer = the one-layer launches the host
                              //   has enqueued before it (an argument) + this
#if defined(GNDT_COST_STAMPS)
    unsigned long long phase[6];   // diagnostic build: cycles wave 0 of the one-workgroup kernel spent per phase of a layer, summed
#endif
};

// While a flood runs, h of a slope that no relaxation has reached yet is this marker (above every float, FLT_MAX included): the
// relaxation that finds it is the slope's first and queues it — what the reference's three-list membership test decides
// (map2D.h:1331-1336: a slope is queued once) — with the atomic min itself, no second flag.  Closed slopes hold FLT_MAX
// (map2D.h:1340), so later relaxations of them never see the marker.  k_cost_finish turns what is left of it into FLT_MAX.
constexpr uint32_t kUnreachedBits = 0x7F800000u;
constexpr uint32_t kFltMaxBits = 0x7F7FFFFFu;

static __global__ void __launch_bounds__(256) k_cost_clear(uint32_t* __restrict__ h_bits, uint32_t* __restrict__ state, uint32_t n,
                                                    uint64_t* __restrict__ ctab_key, uint32_t ctab_size, CostCounters* __restrict__ cc) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint32_t i = gid; i < n; i += gsz) h_bits[i] = kUnreachedBits;
    for (uint32_t i = gid; i < ctab_size; i += gsz) ctab_key[i] = kEmptyKey;       // (ctab_size 0: the column index of the last flood is kept)
    if (gid == 0) {
        cc->frontier[0] = cc->frontier[1] = cc->frontier[2] = 0u;
        cc->traversable = cc->closed = cc->pad0 = cc->range_error = 0u;
        cc->goal_status = 1; cc->levels = 0u; cc->check_pushes = 0ull;
        cc->wg_layers = 0u;
#if defined(GNDT_COST_STAMPS)
        for (int k = 0; k < 6; ++k) cc->phase[k] = 0ull;
#endif
    }
}

// The end of a flood.  What no relaxation reached keeps the FLT_MAX it was created with (map2D.h:636, 652).  state: 1 = the slope was
// expanded (the reference's traversableList), 2 = it collided (closeList), 0 = never queued — every slope whose h left the marker was
// queued, hence popped before the flood ended, and what happened to it then is CollisionCheck's verdict, which self holds for every
// slope (so the layers store no state: a store counts in vmcnt like a load, and a layer is a chain of waits).
static __global__ void __launch_bounds__(256) k_cost_finish(uint32_t* __restrict__ h_bits, uint32_t* __restrict__ state,
                                                     const uint32_t* __restrict__ self, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool reached = h_bits[i] != kUnreachedBits;
        if (!reached) h_bits[i] = kFltMaxBits;
        state[i] = !reached ? 0u : (self && self[2 * (size_t)i + 1]) ? 2u : 1u;
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

// ---------------------------------------------------------------------------------------------
// Per-flood tables: a row's neighbour columns, its own column, and the collision verdict of every slope ("CollisionCheck without
// walking rings" above).
// ---------------------------------------------------------------------------------------------
// nbr / self for every row; and, for floods with rings, the step masks of every slope and round 0 of the extremes
static __global__ void __launch_bounds__(256) k_cost_neighbours(CostView V, Robot R, uint32_t num_rows, int ring_n, uint32_t* __restrict__ nbr,
                                                         uint32_t* __restrict__ self, uint32_t* __restrict__ step, float* __restrict__ hi0,
                                                         float* __restrict__ lo0, CostEdge* __restrict__ edges) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < 4u * num_rows; t += gridDim.x * blockDim.x) {
        const uint32_t row = t >> 2, k = t & 3u;
        uint32_t c, ncol;
        neighbour_column(V, row, k, c, ncol);            // (V.nbr, V.self are null here: everything goes through the hash table)
        nbr[2 * (size_t)t] = c;
        nbr[2 * (size_t)t + 1] = ncol;
        const bool slope = row_has_slope(V, row);
        uint32_t mask = 0u;
        if (slope) {                                       // (one walk over the cell's rows for the record and the ring's steps)
            const CostEdgeAndSteps w = ring_n > 0 ? cost_edge_walk<true>(V, R, row, c, ncol) : cost_edge_walk<false>(V, R, row, c, ncol);
            mask = w.steps;
            reinterpret_cast<uint4*>(edges)[t] = make_uint4(w.e.c, w.e.info, float_bits(w.e.d0), float_bits(w.e.d1));
        }
        if (ring_n > 0) step[t] = mask;
        if (k == 0u) {
            // the row's own column without the hash table: a column's rows are adjacent and its first row is the one that names its size
            uint32_t c_self = row;
            while (V.row_ncol[c_self] == 0u && c_self > 0u) --c_self;
            self[2 * (size_t)row] = c_self;
            const bool up = slope && row_up_in(V, row, c_self);
            self[2 * (size_t)row + 1] = slope && (up || row_above_hits_in(V, R, row, c_self)) ? 1u : 0u;
            if (ring_n > 0 && slope) {
                float hi, lo;
                ring_round0(V, row, up, hi, lo);
                hi0[row] = hi;
                if (V.demand_true) lo0[row] = lo;
            }
        }
    }
}

// one round for every slope, four lanes per slope (one per neighbour cell); the last round adds the verdict to self[2 q + 1]
static __global__ void __launch_bounds__(256) k_cost_ring_round(CostView V, Robot R, uint32_t num_rows, const uint32_t* __restrict__ step,
                                                         const float* __restrict__ hi_in, float* __restrict__ hi_out,
                                                         const float* __restrict__ lo_in, float* __restrict__ lo_out,
                                                         uint32_t* __restrict__ self, int last) {
    for (uint32_t t0 = blockIdx.x * blockDim.x; t0 < 4u * num_rows; t0 += gridDim.x * blockDim.x) {     // (uniform: the quad exchanges below)
        const uint32_t t = t0 + threadIdx.x;
        const bool live = t < 4u * num_rows;
        const uint32_t row = live ? t >> 2 : 0u, k = t & 3u;
        const bool slope = live && row_has_slope(V, row);
        float hi = -FLT_MAX, lo = FLT_MAX;
        if (slope) {
            uint32_t c, ncol;
            neighbour_column(V, row, k, c, ncol);
            ring_round_cell(V, R, row, c, ncol, step[t], hi_in, lo_in, hi, lo);
        }
        hi = fmaxf(hi, __shfl_xor(hi, 1, 64)); hi = fmaxf(hi, __shfl_xor(hi, 2, 64));
        lo = fminf(lo, __shfl_xor(lo, 1, 64)); lo = fminf(lo, __shfl_xor(lo, 2, 64));
        if (slope && k == 0u) {
            hi = fmaxf(hi, hi_in[row]);                // (R_d(q) holds R_(d-1)(q): q itself and everything found so far)
            if (V.demand_true) lo = fminf(lo, lo_in[row]);
            hi_out[row] = hi;
            if (V.demand_true) lo_out[row] = lo;
            if (last && ring_verdict(V, R, row, hi, lo)) self[2 * (size_t)row + 1] = 1u;
        }
    }
}

// goal lookup (map2D.h:1294-1307): the slope of the goal's cell at the goal's level gets h = 0 and is queued
static __global__ void k_cost_goal(CostView V, int gx, int gy, int gz, uint32_t* __restrict__ h_bits, uint32_t* __restrict__ frontier0,
                            CostCounters* __restrict__ cc) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const uint32_t c = ctab_find(V, gx, gy);
    if (c == kNoColumn) { cc->goal_status = 1; return; }
    cc->goal_status = 2;
    const uint32_t b = c, e = c + V.row_ncol[c];
    for (uint32_t t = b; t < e; ++t) {
        if (V.sz[t] == gz && row_has_slope(V, t)) {
            h_bits[t] = 0u;
            frontier0[0] = t;
            cc->frontier[0] = 1u;
            cc->goal_status = 0;
            return;
        }
    }
}

// One layer of the flood: verdict, then expansion, of every slope in the layer.  Four lanes share a slope, one per neighbour cell;
// layers are short, so the work is latency-bound and the dependent round trips per lane count.  Round 5: two of them — what only
// needs the slope (its verdict, h, and the neighbour cell's CostEdge record, requested together), then the atomic minima on the
// accessible slopes' h; the gates and the travel cost (two fp64 square roots, a division, an arc cosine per row of the cell: the
// larger part of a layer's 13-20 k cycles until round 4) were worked out for every slope at once before the flood.  The body is
// shared by the one-layer launch (k_cost_level: every workgroup a wavefront, any layer size) and the one-workgroup kernel that walks
// many layers per launch (k_cost_flood_wg, WG = true: h, which other wavefronts of the workgroup changed one barrier ago, is read
// past the CU's cache; the frontier is in LDS).
#if defined(GNDT_COST_STAMPS)
struct LayerStats { uint32_t trav, closed, checks; unsigned long long ph[6], last; };
#define GNDT_COST_STAMP(k) do { const unsigned long long now_ = clock64(); st.ph[k] += now_ - st.last; st.last = now_; } while (0)
#else
struct LayerStats { uint32_t trav, closed, checks; };
#define GNDT_COST_STAMP(k) do { } while (0)
#endif

// atomic min on h: in global memory, or (LDSH) on the one-workgroup kernel's copy of h in LDS
template <bool LDSH>
__device__ __forceinline__ uint32_t cost_h_min(uint32_t* p, uint32_t v) {
    if constexpr (LDSH)
        return __hip_atomic_fetch_min((__attribute__((address_space(3))) uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        return atomicMin(p, v);
}

// f_in: the layer's slopes (WG: in LDS).  f_out: the next layer's, in global memory — and, WG, its first f_lds_cap entries in LDS
// as well (f_out_lds), where the workgroup's next layer reads them.
// LDSH (WG only): h_bits is the workgroup's copy of h in LDS (maps of up to kCostLdsRows rows): the minima are ds_min_rtn — ~130
// cycles instead of the 1 230-1 560 of a returning global atomic, the longest link of a layer's chain.
template <bool WG, bool LDSH = false>
__device__ __forceinline__ void cost_layer(const CostView& V, const Robot& R, uint32_t n_in, uint32_t first, uint32_t stride,
                                           uint32_t* __restrict__ h_bits, uint32_t* __restrict__ state,
                                           const uint32_t* f_in, uint32_t* __restrict__ f_out, uint32_t* f_out_lds, uint32_t f_lds_cap,
                                           uint32_t* out_count, LayerStats& st, uint32_t* dump = nullptr) {
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t dir = threadIdx.x & 3u;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    // dump (LDSH): 64 words of LDS nobody reads.  A word of the record and of the verdict of every slope a lane queues is requested
    // into it (LDS-DMA: no register waits for the data), so that the lines are on their way to the CU when the next layer asks.
    auto touch = [&](uint32_t t) {
        if constexpr (LDSH) {
            if (dump) {
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(V.edges) + 16 * (size_t)t, (__attribute__((address_space(3))) uint32_t*)dump, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(V.self + 2 * (size_t)t + 1, (__attribute__((address_space(3))) uint32_t*)dump, 4, 0, 0);
            }
        }
    };
    // The loop is wave-uniform (a wave's quads take 16 consecutive slopes of the layer, lanes past the end sit idle), so that the
    // slopes a wave pushes are appended with ONE atomic on the layer's counter: the counter is one word, same-address atomics
    // retire at ~90 per microsecond at the memory side, and a layer of a few hundred slopes used to add one per pushed slope
    // and three more per expanded slope (the statistics) — most of a layer's 13-18 us in round 3.
    for (uint32_t i0 = first; i0 < n_in; i0 += stride) {
        const uint32_t i = i0 + ((uint32_t)lane >> 2);
        const bool live = i < n_in;
        const uint32_t q = live ? f_in[i] : 0u;
        GNDT_COST_STAMP(0);
        uint32_t hq_bits = 0u, hit = 0u;
        uint4 er = make_uint4(kNoColumn, 0u, 0u, 0u);
        if (live) {
            hit = V.self[2 * (size_t)q + 1];           // CollisionCheck's verdict, found for every slope before the flood
            hq_bits = LDSH ? h_bits[q] : WG ? __hip_atomic_load(&h_bits[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : h_bits[q];
            er = reinterpret_cast<const uint4*>(V.edges)[4 * (size_t)q + dir];
            // (all three in flight together: left alone, the compiler asks for the record only once the verdict is back)
            asm volatile("" : "+v"(hit), "+v"(hq_bits), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er.w));
        }
        GNDT_COST_STAMP(1);
        uint32_t p0 = kNone, p1 = kNone;               // the slopes this lane queues (a third and later ones go out one by one)
        bool closed_one = false;
        if (live) {
            if (hit) {
                if (dir == 0u) {
                    h_bits[q] = kFltMaxBits;          // Q.front()->h = FLT_MAX (map2D.h:1340)
                    ++st.closed;
                    closed_one = true;
                }
            } else {
                if (dir == 0u) ++st.trav;
                const float hq = bits_float(hq_bits);
                if (!(er.y & kEdgeMore)) {
                    // cost_expand_record, both minima in flight together
                    const uint32_t n = (er.y >> 16) & 3u;
                    const uint32_t t0 = er.x + (er.y & 0xFFu), t1 = er.x + ((er.y >> 8) & 0xFFu);
                    const uint32_t c0 = float_bits(hq + bits_float(er.z)), c1 = float_bits(hq + bits_float(er.w));
                    // (a candidate not below the FLT_MAX every h starts from is no improvement, map2D.h:1329)
                    const bool do0 = n > 0u && c0 < kFltMaxBits, do1 = n > 1u && c1 < kFltMaxBits;
                    uint32_t o0 = 0u, o1 = 0u;
                    if (do0) o0 = cost_h_min<LDSH>(&h_bits[t0], c0);
                    if (do1) o1 = cost_h_min<LDSH>(&h_bits[t1], c1);
                    asm volatile("" : "+v"(o0), "+v"(o1));             // (one wait for both, not one after each)
                    if (do0 && o0 == kUnreachedBits) p0 = t0;          // (the slope's first relaxation queues it)
                    if (do1 && o1 == kUnreachedBits) p1 = t1;
                    st.checks += er.y >> 20;
                } else {
                    float nq[3], mq[3];
                    for (int k = 0; k < 3; ++k) { nq[k] = V.normal[3 * (size_t)q + k]; mq[k] = V.mean[3 * (size_t)q + k]; }
                    uint32_t nc, nc_rows;
                    neighbour_column(V, q, dir, nc, nc_rows);
                    st.checks += cost_expand_column(V, R, hq, nq, mq, nc, nc_rows, [&](uint32_t t, float cand) {
                        const uint32_t cb = float_bits(cand);
                        if (cb >= kFltMaxBits) return;
                        if (cost_h_min<LDSH>(&h_bits[t], cb) != kUnreachedBits) return;
                        if (p0 == kNone) p0 = t;
                        else if (p1 == kNone) p1 = t;
                        else {
                            const uint32_t pos = atomicAdd(out_count, 1u);
                            f_out[pos] = t;
                            if (WG && pos < f_lds_cap) f_out_lds[pos] = t;
                        }
                    });
                }
            }
        }
        GNDT_COST_STAMP(2);
        const unsigned long long b0 = __ballot(p0 != kNone), b1 = __ballot(p1 != kNone);
        if (b0 | b1) {                                 // (wave-uniform)
            const uint32_t n0 = (uint32_t)__popcll(b0), total = n0 + (uint32_t)__popcll(b1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(out_count, total);
            base = (uint32_t)__shfl((int)base, 0, 64);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (p0 != kNone) {
                const uint32_t pos = base + (uint32_t)__popcll(b0 & below);
                f_out[pos] = p0;
                if (WG && pos < f_lds_cap) f_out_lds[pos] = p0;
                touch(p0);
            }
            if (p1 != kNone) {
                const uint32_t pos = base + n0 + (uint32_t)__popcll(b1 & below);
                f_out[pos] = p1;
                if (WG && pos < f_lds_cap) f_out_lds[pos] = p1;
                touch(p1);
            }
        }
        // WG: the FLT_MAX of a closed slope must have arrived before another wavefront's atomic min on the same word, one barrier on
        if (WG && !LDSH && __any(closed_one)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (LDSH: the layer's barrier waits for LDS)
        GNDT_COST_STAMP(3);
    }
}

// the statistics of a wave: one atomic per counter
__device__ __forceinline__ void cost_flush_stats(LayerStats st, CostCounters* __restrict__ cc) {
    for (int o = 32; o > 0; o >>= 1) {
        st.trav += (uint32_t)__shfl_down((int)st.trav, o, 64); st.closed += (uint32_t)__shfl_down((int)st.closed, o, 64);
        st.checks += (uint32_t)__shfl_down((int)st.checks, o, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        if (st.trav) atomicAdd(&cc->traversable, st.trav);
        if (st.closed) atomicAdd(&cc->closed, st.closed);
        if (st.checks) atomicAdd(&cc->check_pushes, (unsigned long long)st.checks);
    }
}

// ONE layer, any size: a launch of single-wavefront workgroups.  The layer's number is `launched` — the one-layer launches enqueued
// before this one — plus the layers the one-workgroup kernel has walked (cc->wg_layers; the host does not know how far that got).
static __global__ void __launch_bounds__(64) k_cost_level(CostView V, Robot R, uint32_t* __restrict__ h_bits, uint32_t* __restrict__ state,
                                                   uint32_t* __restrict__ f0, uint32_t* __restrict__ f1, CostCounters* __restrict__ cc,
                                                   uint32_t launched) {
    const uint32_t level = launched + cc->wg_layers;
    const uint32_t n_in = cc->frontier[level % 3u];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cc->frontier[(level + 2u) % 3u] = 0u;          // the counter two layers ahead (was the previous layer's input)
        if (n_in) cc->levels = level + 1u;
    }
    if (blockIdx.x * 16u >= n_in) return;              // (uniform; every workgroup once the flood has ended)
    LayerStats st{};
    cost_layer<false>(V, R, n_in, blockIdx.x * 16u, gridDim.x * 16u, h_bits, state, (level & 1u) ? f1 : f0, (level & 1u) ? f0 : f1, nullptr, 0u,
                      &cc->frontier[(level + 1u) % 3u], st);
    cost_flush_stats(st, cc);
}

// MANY layers per launch, while they are narrow: one workgroup of 16 wavefronts walks layer after layer with ONE workgroup barrier
// between them instead of a launch, the frontier and its counters in LDS.  It stops at a layer wider than max_frontier (<= kWgFrontier;
// the one-layer launches that follow it on the stream take it from there: cc->wg_layers, cc->frontier, the frontier arrays in global
// memory are kept complete), when the flood has ended, or after max_layers.
// Measured (profiles/r05_cost_map.json): 2.9 us per layer on the site, 2.3 on the 8 M-point terrain (round 4, with the gates and the
// travel cost inside the layers: 6.5 / 7.9); one-layer launches 5.9 / 5.9.  Where a layer's ~5 k cycles go (tools/cost_stamps.sh,
// profiles/r05_cost_stamps.txt): its two returning round trips — the loads ~600, the atomic minima ~1 300 (tools/atomic_latency.hip:
// a returning global atomic takes 1 230-1 560 cycles on an idle chip, a load 370-880) — and the barrier, i.e. the slowest wave's.
constexpr int kWgThreads = 1024;
constexpr uint32_t kWgFrontier = 4u * (uint32_t)(kWgThreads / 4);      // four rounds of the workgroup's quads

// LDSH: h of the WHOLE map in LDS for the time of the launch (n_rows <= kCostLdsRows: 144 KB of the CU's 160), read from h_bits at the
// start and written back at the end — whatever runs behind this kernel on the stream finds h where it always is.
constexpr uint32_t kCostLdsRows = 36u * 1024u;

template <bool LDSH>
static __global__ void __launch_bounds__(kWgThreads) k_cost_flood_wg(CostView V, Robot R, uint32_t* __restrict__ h_bits,
                                                              uint32_t* __restrict__ state, uint32_t* f0, uint32_t* f1,
                                                              CostCounters* __restrict__ cc, uint32_t max_frontier, uint32_t max_layers,
                                                              uint32_t launched, uint32_t n_rows, int prefetch) {
    __shared__ uint32_t s_count[3];
    __shared__ uint32_t s_dump[64];
    __shared__ uint32_t s_f[2][kWgFrontier];
    extern __shared__ uint32_t s_h[];                  // (LDSH: n_rows words)
    uint32_t level = launched + cc->wg_layers;
    uint32_t n_in = cc->frontier[level % 3u];
    max_frontier = min(max_frontier, kWgFrontier);
    if (n_in == 0u || n_in > max_frontier) return;     // (uniform) nothing for this kernel to do: cc stays as it is
    {
        const uint32_t* f_first = (level & 1u) ? f1 : f0;
        for (uint32_t j = threadIdx.x; j < n_in; j += blockDim.x) s_f[level & 1u][j] = f_first[j];
        if (threadIdx.x < 3u) s_count[threadIdx.x] = 0u;
        if constexpr (LDSH) for (uint32_t j = threadIdx.x; j < n_rows; j += blockDim.x) s_h[j] = h_bits[j];
    }
    __syncthreads();
    uint32_t* const h_live = LDSH ? s_h : h_bits;
    LayerStats st{};
#if defined(GNDT_COST_STAMPS)
    st.last = clock64();
#endif
    uint32_t done = 0;
    for (; done < max_layers && n_in != 0u && n_in <= max_frontier; ++done) {
        if (threadIdx.x == 0) s_count[(level + 2u) % 3u] = 0u;        // (the layer after next's; nobody looks at it during this layer)
        cost_layer<true, LDSH>(V, R, n_in, (threadIdx.x >> 6) * 16u, (uint32_t)kWgThreads / 4u, h_live, state, s_f[level & 1u], (level & 1u) ? f0 : f1,
                               s_f[(level + 1u) & 1u], kWgFrontier, &s_count[(level + 1u) % 3u], st, prefetch ? s_dump : nullptr);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS only: the frontier's global copy is for later kernels)
        ++level;
        n_in = s_count[level % 3u];
        GNDT_COST_STAMP(4);
    }
#if defined(GNDT_COST_STAMPS)
    if (threadIdx.x == 0) { for (int k = 0; k < 5; ++k) cc->phase[k] += st.ph[k]; cc->phase[5] += done; }
#endif
    if constexpr (LDSH) for (uint32_t j = threadIdx.x; j < n_rows; j += blockDim.x) h_bits[j] = s_h[j];
    cost_flush_stats(st, cc);
    if (threadIdx.x == 0) {
        cc->wg_layers = level - launched;
        cc->frontier[level % 3u] = n_in;
        cc->frontier[(level + 1u) % 3u] = 0u;
        cc->frontier[(level + 2u) % 3u] = 0u;
        if (done) cc->levels = level;                  // (every layer walked here held at least one slope)
    }
}
#endif  // __HIPCC__

}  // namespace gndt
