// Test-only host shim: exposes grid_ndt_amd/csrc/gndt_math.hpp (the arithmetic the HIP kernels run)
// to the CPU test tier, so key computation, finalisation and the eigen-solve can be checked against
// the oracle without a GPU.  Not part of the product library.
#include <stdint.h>
#include "gndt_math.hpp"
extern "C" {
void shim_point_keys(const float* xyz, uint64_t n, int stride, const float o[3], float gl, float zl,
                     uint64_t* keys, uint8_t* ok) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = xyz + i * stride;
        gndt::PointKey k = gndt::point_key(p[0], p[1], p[2], o[0], o[1], o[2], gl, zl);
        keys[i] = gndt::pack_key(k.sx, k.sy, k.sz);
        ok[i] = k.ok;
    }
}
// the divide-free form the hot kernels use (axis_index_fast): must agree with shim_point_keys bit for bit
void shim_point_keys_fast(const float* xyz, uint64_t n, int stride, const float o[3], float gl, float zl,
                          uint64_t* keys, uint8_t* ok) {
    const float ig = 1.0f / gl, iz = 1.0f / zl;
    for (uint64_t i = 0; i < n; ++i) {
        const float* p = xyz + i * stride;
        gndt::PointKey k = gndt::point_key_fast(p[0], p[1], p[2], o[0], o[1], o[2], gl, zl, ig, iz);
        keys[i] = gndt::pack_key(k.sx, k.sy, k.sz);
        ok[i] = k.ok;
    }
}
void shim_centres(const uint64_t* keys, uint64_t n, const float o[3], float gl, float zl, double* c) {
    for (uint64_t i = 0; i < n; ++i) {
        int sx, sy, sz;
        gndt::unpack_key(keys[i], sx, sy, sz);
        c[3 * i] = gndt::axis_centre(sx, o[0], gl);
        c[3 * i + 1] = gndt::axis_centre(sy, o[1], gl);
        c[3 * i + 2] = gndt::axis_centre(sz, o[2], zl);
    }
}
void shim_finalize(const uint32_t* count, const double* sums, const double* centres, uint64_t n, int min_points,
                   float* mean, float* cov, float* rough, float* normal) {
    for (uint64_t i = 0; i < n; ++i) {
        gndt::NodeResult r{};
        if ((int)count[i] >= min_points) gndt::finalize_node(count[i], sums + 9 * i, centres + 3 * i, r);
        for (int k = 0; k < 3; ++k) { mean[3 * i + k] = r.mean[k]; normal[3 * i + k] = r.normal[k]; }
        for (int k = 0; k < 6; ++k) cov[6 * i + k] = r.cov[k];
        rough[i] = r.rough;
    }
}
float shim_mean_z(uint32_t n, double sum_vz, double cz) { return gndt::node_mean_z(n, sum_vz, cz); }
int shim_level_above(int z) { return gndt::level_above(z); }
int shim_level_below(int z) { return gndt::level_below(z); }
}
extern "C" void shim_min_eigen(const double* S, uint64_t n, double* lam, double* vec) {
    for (uint64_t i = 0; i < n; ++i) gndt::min_eigenpair_sym3(S + 6 * i, lam[i], vec + 3 * i);
}
extern "C" void shim_jacobi(const double* S, uint64_t n, double* evals, double* evecs) {
    for (uint64_t i = 0; i < n; ++i) {
        double ev[3], vv[3][3];
        gndt::eigen_sym3(S + 6 * i, ev, vv);
        for (int k = 0; k < 3; ++k) { evals[3 * i + k] = ev[k]; for (int j = 0; j < 3; ++j) evecs[9 * i + 3 * k + j] = vv[k][j]; }
    }
}

// ---- cost-map flood (grid_ndt_amd/csrc/gndt_cost.hpp): the per-slope logic the kernels run, driven level by
// level on the host exactly as k_cost_level is launched layer after layer: a slope's four CostEdge records (what k_cost_neighbours
// writes for every slope before the flood), then the expansion from them ----
#include <vector>
#include "gndt_cost.hpp"
extern "C" int shim_cost(uint64_t n, const int32_t* sx, const int32_t* sy, const int32_t* sz, const float* mean,
                         const float* normal, const float* rough, const uint32_t* flags, float slope_interval,
                         int demand_true, float grid_len, int gx, int gy, int gz, const float robot4[4], float* h_out,
                         uint8_t* state_out, int64_t stats[8]) {
    using namespace gndt;
    // columns are contiguous in the reference order
    std::vector<uint32_t> col_base, row_ncol(n ? n : 1, 0u);
    for (uint64_t i = 0; i < n; ++i) {
        if (i == 0 || sx[i] != sx[i - 1] || sy[i] != sy[i - 1]) col_base.push_back((uint32_t)i);
        ++row_ncol[col_base.back()];
    }
    uint32_t tsize = 1024;
    while (tsize < 2 * col_base.size()) tsize <<= 1;
    std::vector<uint64_t> tkey(tsize, kEmptyKey);
    std::vector<uint32_t> tval(tsize, 0);
    for (uint32_t c = 0; c < col_base.size(); ++c) {
        const uint64_t key = column_pack(sx[col_base[c]], sy[col_base[c]]);
        uint32_t s = (uint32_t)mix64(key) & (tsize - 1);
        while (tkey[s] != kEmptyKey) s = (s + 1) & (tsize - 1);
        tkey[s] = key; tval[s] = col_base[c];
    }
    CostView V{sx, sy, sz, mean, normal, rough, flags, row_ncol.data(), tkey.data(), tval.data(), tsize - 1, nullptr,
               slope_interval, demand_true};
    Robot R{robot4[0], robot4[1], robot4[2], robot4[3]};
    const int ring_n = cost_ring_depth(R.r, grid_len);
    std::vector<uint32_t> hb(n, 0x7F7FFFFFu), pushed(n, 0), state(n, 0), frontier, next, ring(kRingCap);
    int64_t trav = 0, closed = 0, checks = 0, levels = 0, overflow = 0, records = 0, records_more = 0;
    int goal_status = 1;
    const uint32_t gc = ctab_find(V, gx, gy);
    if (gc != kNoColumn) {
        goal_status = 2;
        for (uint32_t t = gc; t < gc + row_ncol[gc]; ++t)
            if (sz[t] == gz && row_has_slope(V, t)) { hb[t] = 0; pushed[t] = 1; frontier.push_back(t); goal_status = 0; break; }
    }
    while (!frontier.empty()) {
        ++levels;
        next.clear();
        for (uint32_t q : frontier) {
            const int hit = cost_collide(V, R, q, ring_n, ring.data());
            if (hit < 0) { ++overflow; continue; }
            if (hit) { hb[q] = 0x7F7FFFFFu; state[q] = 2; ++closed; continue; }
            state[q] = 1; ++trav;
            auto relax = [&](uint32_t t, float cand) {
                const uint32_t cb = float_bits(cand);
                const uint32_t old = hb[t];
                if (cb < old) hb[t] = cb;
                if (old > cb && pushed[t] == 0) { pushed[t] = 1; next.push_back(t); }
            };
            for (uint32_t k = 0; k < 4u; ++k) {          // as cost_layer: the record, or the column itself where the record says so
                uint32_t c, ncol;
                neighbour_column(V, q, k, c, ncol);
                const CostEdge e = cost_edge_record(V, R, q, c, ncol);
                ++records;
                if (e.info & kEdgeMore) {
                    ++records_more;
                    checks += cost_expand_column(V, R, bits_float(hb[q]), V.normal + 3 * (size_t)q, V.mean + 3 * (size_t)q, c, ncol, relax);
                } else {
                    checks += cost_expand_record(e, bits_float(hb[q]), relax);
                }
            }
        }
        frontier.swap(next);
    }
    for (uint64_t i = 0; i < n; ++i) { h_out[i] = bits_float(hb[i]); state_out[i] = (uint8_t)state[i]; }
    stats[0] = trav; stats[1] = closed; stats[2] = checks; stats[3] = ring_n; stats[4] = levels; stats[5] = overflow;
    stats[6] = records; stats[7] = records_more;
    return goal_status;
}


// CollisionCheck for EVERY slope of a grid, twice: the walk over the ring (cost_collide, the restatement of map2D.h:351-474) and the
// rounds over the whole map the device runs instead (gndt_cost.hpp, "CollisionCheck without walking rings").  walk / rounds: 1 collide,
// 0 free, 255 not a slope (walk: 2 = the walk's ring scratch of ring_cap slopes did not fit).
extern "C" int shim_collide_all(uint64_t n, const int32_t* sx, const int32_t* sy, const int32_t* sz, const float* mean,
                                const float* normal, const float* rough, const uint32_t* flags, float slope_interval,
                                int demand_true, float grid_len, const float robot4[4], int ring_cap, uint8_t* walk, uint8_t* rounds) {
    using namespace gndt;
    std::vector<uint32_t> col_base, row_ncol(n ? n : 1, 0u);
    for (uint64_t i = 0; i < n; ++i) {
        if (i == 0 || sx[i] != sx[i - 1] || sy[i] != sy[i - 1]) col_base.push_back((uint32_t)i);
        ++row_ncol[col_base.back()];
    }
    uint32_t tsize = 1024;
    while (tsize < 2 * col_base.size()) tsize <<= 1;
    std::vector<uint64_t> tkey(tsize, kEmptyKey);
    std::vector<uint32_t> tval(tsize, 0);
    for (uint32_t c = 0; c < col_base.size(); ++c) {
        const uint64_t key = column_pack(sx[col_base[c]], sy[col_base[c]]);
        uint32_t s = (uint32_t)mix64(key) & (tsize - 1);
        while (tkey[s] != kEmptyKey) s = (s + 1) & (tsize - 1);
        tkey[s] = key; tval[s] = col_base[c];
    }
    CostView V{sx, sy, sz, mean, normal, rough, flags, row_ncol.data(), tkey.data(), tval.data(), tsize - 1, nullptr,
               slope_interval, demand_true};
    Robot R{robot4[0], robot4[1], robot4[2], robot4[3]};
    const int ring_n = cost_ring_depth(R.r, grid_len);
    std::vector<uint32_t> ring((size_t)ring_cap);
    for (uint64_t q = 0; q < n; ++q) {
        if (!row_has_slope(V, (uint32_t)q)) { walk[q] = 255; continue; }
        const int hit = cost_collide(V, R, (uint32_t)q, ring_n, ring.data(), ring_cap);
        walk[q] = hit < 0 ? 2 : (uint8_t)hit;
    }
    // the rounds, as k_cost_neighbours / k_cost_ring_round run them
    std::vector<uint32_t> nc(4 * n), nr(4 * n), step(4 * n, 0u);
    std::vector<float> hi[2] = {std::vector<float>(n, 0.f), std::vector<float>(n, 0.f)}, lo[2] = {std::vector<float>(n, 0.f), std::vector<float>(n, 0.f)};
    std::vector<uint8_t> verdict(n, 0);
    std::vector<uint8_t> differs(n, 0);      // the kernel's shortcuts against the plain statements (a slope where they differ reads 3 below)
    for (uint64_t q = 0; q < n; ++q) {
        const bool slope = row_has_slope(V, (uint32_t)q);
        for (uint32_t k = 0; k < 4; ++k) {
            neighbour_column(V, (uint32_t)q, k, nc[4 * q + k], nr[4 * q + k]);
            if (ring_n > 0 && slope) {
                // k_cost_neighbours: the ring's steps come out of the walk that makes the CostEdge record
                step[4 * q + k] = cost_edge_walk<true>(V, R, (uint32_t)q, nc[4 * q + k], nr[4 * q + k]).steps;
                if (step[4 * q + k] != ring_step_mask(V, R, (uint32_t)q, nc[4 * q + k], nr[4 * q + k])) differs[q] = 1;
            }
        }
        // k_cost_neighbours: the row's own column by walking back to the row that names the column's size, not through the hash table
        uint32_t c_self = (uint32_t)q;
        while (row_ncol[c_self] == 0u && c_self > 0u) --c_self;
        if (c_self != ctab_find(V, sx[q], sy[q])) differs[q] = 1;
        const bool up = slope && row_up_in(V, (uint32_t)q, c_self);
        if (up != (slope && row_up(V, (uint32_t)q))) differs[q] = 1;
        verdict[q] = slope && (up || row_above_hits_in(V, R, (uint32_t)q, c_self)) ? 1 : 0;
        if (verdict[q] != (slope && ring_free_verdict(V, R, (uint32_t)q, up) ? 1 : 0)) differs[q] = 1;
        if (ring_n > 0 && slope) ring_round0(V, (uint32_t)q, up, hi[0][q], lo[0][q]);
    }
    for (int d = 0; d < ring_n; ++d) {
        const std::vector<float>&hin = hi[d & 1], &lin = lo[d & 1];
        std::vector<float>&hout = hi[(d + 1) & 1], &lout = lo[(d + 1) & 1];
        for (uint64_t q = 0; q < n; ++q) {
            if (!row_has_slope(V, (uint32_t)q)) continue;
            float h = hin[q], l = lin[q];
            for (uint32_t k = 0; k < 4; ++k) ring_round_cell(V, R, (uint32_t)q, nc[4 * q + k], nr[4 * q + k], step[4 * q + k], hin.data(), lin.data(), h, l);
            hout[q] = h; lout[q] = l;
            if (d == ring_n - 1 && ring_verdict(V, R, (uint32_t)q, h, l)) verdict[q] = 1;
        }
    }
    for (uint64_t q = 0; q < n; ++q) rounds[q] = differs[q] ? 3 : row_has_slope(V, (uint32_t)q) ? verdict[q] : 255;
    return ring_n;
}
