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
