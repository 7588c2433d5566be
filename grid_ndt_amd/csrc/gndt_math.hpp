// gndt_math.hpp — per-point and per-node arithmetic shared by every kernel (and callable on the host
// so the CPU-only test tier can exercise it).  No reference code is reused; each routine cites the
// reference statement whose RESULT it must reproduce.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GNDT_HD __host__ __device__ __forceinline__
#else
#define GNDT_HD inline
#endif

namespace gndt {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kMaxXY = 65535;          // countMorton is unique only up to here (Stopwatch.h:102-110)
constexpr int kMaxZ = (1 << 21) - 1;   // 22-bit biased field

// ---- packed node key: bits 63..43 sx+2^20 | 42..22 sy+2^20 | 21..0 sz+2^21 ----------------------
GNDT_HD uint64_t pack_key(int sx, int sy, int sz) {
    return ((uint64_t)((uint32_t)(sx + (1 << 20)) & 0x1FFFFFu) << 43) |
           ((uint64_t)((uint32_t)(sy + (1 << 20)) & 0x1FFFFFu) << 22) |
           ((uint64_t)((uint32_t)(sz + (1 << 21)) & 0x3FFFFFu));
}
GNDT_HD void unpack_key(uint64_t k, int& sx, int& sy, int& sz) {
    sx = (int)((k >> 43) & 0x1FFFFFu) - (1 << 20);
    sy = (int)((k >> 22) & 0x1FFFFFu) - (1 << 20);
    sz = (int)(k & 0x3FFFFFu) - (1 << 21);
}
GNDT_HD uint64_t column_key(uint64_t node_key) { return node_key & ~0x3FFFFFull; }  // sz field = 0 (never a node)

GNDT_HD uint64_t mix64(uint64_t x) {  // murmur3 finaliser
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// ---- transMortonXYZ in integer form (include/map2D.h:950-976) -----------------------------------
// One axis: n = (int)ceilf(fabsf(p - o) / len), 0 -> 1; sign + iff p > o (strict).
// Must stay IEEE fp32: subtract, abs, correctly rounded divide, ceil.  No reciprocal, no FMA.
GNDT_HD int axis_index(float p, float o, float len, bool& ok, int limit) {
    float d = p - o;
    float q = fabsf(d) / len;
    float c = ceilf(q);
    // c is a non-negative integer-valued float (or inf/nan for bad input); clamp before the cast
    ok = ok && (c <= (float)limit);
    int n = (c <= (float)limit) ? (int)c : limit;
    if (n == 0) n = 1;
    return (p > o) ? n : -n;
}

struct PointKey {
    int sx, sy, sz;
    bool ok;
};

GNDT_HD PointKey point_key(float px, float py, float pz, float ox, float oy, float oz, float grid_len, float z_len) {
    PointKey k;
    k.ok = true;
    k.sx = axis_index(px, ox, grid_len, k.ok, kMaxXY);
    k.sy = axis_index(py, oy, grid_len, k.ok, kMaxXY);
    k.sz = axis_index(pz, oz, z_len, k.ok, kMaxZ);
    return k;
}

// Centre of a node along one axis, in fp64: o + sign*(n - 1/2)*len.  Any fixed function of the key
// would do (the statistics are shift-invariant); the centre keeps |v| <= len/2.
GNDT_HD double axis_centre(int s, float o, float len) {
    double half = (s > 0) ? ((double)s - 0.5) : ((double)s + 0.5);
    return (double)o + half * (double)len;
}

// ---- symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64) ------------------------------------
// Replaces Eigen::EigenSolver<Matrix3f> at include/map2D.h:111-113.  c = xx,xy,xz,yy,yz,zz.
GNDT_HD void jacobi_rotate(double a[3][3], double v[3][3], int p, int q) {
    if (a[p][q] == 0.0) return;
    double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
    double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
    int r = 3 - p - q;  // the untouched index
    double app = a[p][p], aqq = a[q][q], apq = a[p][q];
    a[p][p] = app - t * apq;
    a[q][q] = aqq + t * apq;
    a[p][q] = a[q][p] = 0.0;
    double arp = a[r][p], arq = a[r][q];
    a[r][p] = a[p][r] = cs * arp - sn * arq;
    a[r][q] = a[q][r] = sn * arp + cs * arq;
    for (int k = 0; k < 3; ++k) {
        double vkp = v[k][p], vkq = v[k][q];
        v[k][p] = cs * vkp - sn * vkq;
        v[k][q] = sn * vkp + cs * vkq;
    }
}

GNDT_HD void eigen_sym3(const double c[6], double evals[3], double evecs[3][3]) {
    double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * dia) break;
        jacobi_rotate(a, v, 0, 1);
        jacobi_rotate(a, v, 0, 2);
        jacobi_rotate(a, v, 1, 2);
    }
    for (int i = 0; i < 3; ++i) {
        evals[i] = a[i][i];
        for (int k = 0; k < 3; ++k) evecs[k][i] = v[k][i];
    }
}

// OcNode::countRoughNormal's choice (include/map2D.h:114-130): strict '<', ties to the higher index.
GNDT_HD int pick_min_eigen(const double e[3]) {
    if (e[0] < e[1]) return (e[0] < e[2]) ? 0 : 2;
    return (e[1] < e[2]) ? 1 : 2;
}

// ---- per-node finalisation -----------------------------------------------------------------------
// From additive cell-local statistics (n, Sum v, Sum v v^T) to what OcNode / Slope hold:
//   mean  = centre + Sum v / n                       (pcl::compute3DCentroid,  map2D.h:621)
//   S     = Sum v v^T - (Sum v)(Sum v)^T / n          (pcl::computeCovarianceMatrix, :622; NOT / n)
//   rough = min eigenvalue (0 -> 0.01), normal = its eigenvector (map2D.h:110-133)
struct NodeResult {
    float mean[3];
    float cov[6];
    float rough;
    float normal[3];
};

GNDT_HD void finalize_node(uint32_t n, const double sums[9], const double centre[3], NodeResult& r) {
    const double inv = 1.0 / (double)n;
    double m[3] = {sums[0] * inv, sums[1] * inv, sums[2] * inv};
    double S[6];
    S[0] = sums[3] - sums[0] * m[0];
    S[1] = sums[4] - sums[0] * m[1];
    S[2] = sums[5] - sums[0] * m[2];
    S[3] = sums[6] - sums[1] * m[1];
    S[4] = sums[7] - sums[1] * m[2];
    S[5] = sums[8] - sums[2] * m[2];
    // a scatter matrix is positive semi-definite; cancellation can leave a diagonal at -1e-17
    if (S[0] < 0.0) S[0] = 0.0;
    if (S[3] < 0.0) S[3] = 0.0;
    if (S[5] < 0.0) S[5] = 0.0;
    for (int k = 0; k < 3; ++k) r.mean[k] = (float)(centre[k] + m[k]);
    for (int k = 0; k < 6; ++k) r.cov[k] = (float)S[k];
    double ev[3], vec[3][3];
    eigen_sym3(S, ev, vec);
    int j = pick_min_eigen(ev);
    double nx = vec[0][j], ny = vec[1][j], nz = vec[2][j];
    double nn = sqrt(nx * nx + ny * ny + nz * nz);
    if (nn > 0.0) { nx /= nn; ny /= nn; nz /= nn; }
    // sign is unspecified in the reference (consumers fold the angle, map2D.h:477-482): point it up
    if (nz < 0.0 || (nz == 0.0 && (ny < 0.0 || (ny == 0.0 && nx < 0.0)))) { nx = -nx; ny = -ny; nz = -nz; }
    float rough = (float)ev[j];
    if (rough < 0.f) rough = 0.f;
    if (rough == 0.f) rough = 0.01f;   // map2D.h:131-132
    r.rough = rough;
    r.normal[0] = (float)nx; r.normal[1] = (float)ny; r.normal[2] = (float)nz;
}

// mean z as the reference stores it (fp32), used by the slope test
GNDT_HD float node_mean_z(uint32_t n, double sum_vz, double centre_z) {
    return (float)(centre_z + sum_vz / (double)n);
}

// The zadd / zminus rule of OcNode::isSlope (include/map2D.h:69-75): there is no level 0.
GNDT_HD int level_above(int z) { return (z == -1) ? 1 : z + 1; }
GNDT_HD int level_below(int z) { return (z == 1) ? -1 : z - 1; }

}  // namespace gndt
