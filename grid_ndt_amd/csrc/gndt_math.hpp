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

// The same index without the divide whenever that is PROVABLY the same.  inv_len = RN(1/len) (host, IEEE).
//   q~ = RN(a * inv_len) carries at most 1.5 * 2^-23 relative error against the exact quotient Q, the reference's
//   q = RN(Q) at most 2^-24, so |q~ - q| < 1.8e-7 * Q.  If no integer lies within 2.4e-7 * q~ of q~, none lies
//   between q~ and q (nor on q), hence ceilf(q~) == ceilf(q).  Otherwise (q~ within ~2 ulp of an integer, or not
//   finite) the IEEE divide decides, exactly as axis_index does.  About one axis value in
//   10^4 takes the slow branch at the index magnitudes of the benchmark scenes; results are bit-identical always
//   (tests/test_host_math.py drives both forms over lattice-adversarial inputs).
// One axis of that: ceil of the divide-free quotient, and whether it is undecided.  Spelled with two products instead of
// distances: q~ (1 - 3e-7) and q~ (1 + 3e-7) bracket the reference's quotient (3e-7 leaves room for their own rounding),
// so if both have the same ceiling, that is the ceiling.  NaN compares unequal -> undecided -> the divide reports it.
GNDT_HD float axis_ceil_try(float p, float o, float inv_len, bool& undecided) {
    const float q = fabsf(p - o) * inv_len;
    const float c_hi = ceilf(q * 1.0000003f), c_lo = ceilf(q * 0.9999997f);
    undecided = undecided || !(c_hi == c_lo);
    return c_hi;
}
// c = ceilf(|p - o| / len) however obtained -> signed index (the tail of axis_index)
GNDT_HD int axis_from_ceil(float c, float p, float o, bool& ok, int limit) {
    ok = ok && (c <= (float)limit);
    int n = (int)fminf(c, (float)limit);         // (NaN -> limit; ok is already false)
    if (n == 0) n = 1;
    return (p > o) ? n : -n;
}
GNDT_HD int axis_index_fast(float p, float o, float len, float inv_len, bool& ok, int limit) {
    bool und = false;
    float c = axis_ceil_try(p, o, inv_len, und);
    if (und) c = ceilf(fabsf(p - o) / len);      // undecided: the reference's own arithmetic
    return axis_from_ceil(c, p, o, ok, limit);
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

// All three axes with ONE (rare) branch: if any axis is undecided, all three take the IEEE divide.
GNDT_HD PointKey point_key_fast(float px, float py, float pz, float ox, float oy, float oz, float grid_len, float z_len,
                                float inv_grid, float inv_z) {
    bool und = false;
    float cx = axis_ceil_try(px, ox, inv_grid, und);
    float cy = axis_ceil_try(py, oy, inv_grid, und);
    float cz = axis_ceil_try(pz, oz, inv_z, und);
    if (und) {
        cx = ceilf(fabsf(px - ox) / grid_len);
        cy = ceilf(fabsf(py - oy) / grid_len);
        cz = ceilf(fabsf(pz - oz) / z_len);
    }
    PointKey k;
    k.ok = true;
    k.sx = axis_from_ceil(cx, px, ox, k.ok, kMaxXY);
    k.sy = axis_from_ceil(cy, py, oy, k.ok, kMaxXY);
    k.sz = axis_from_ceil(cz, pz, oz, k.ok, kMaxZ);
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

// ---- minimum eigenpair of a symmetric positive semi-definite 3x3, fast path ------------------------
// What OcNode::countRoughNormal needs (map2D.h:110-133) is only the smallest eigenvalue and its
// eigenvector.  Jacobi (above, kept as the in-repo cross-check) costs thousands of fp64 instructions per
// node; this costs ~200:
//   1. scale by max |S_ij|;
//   2. fp32 trigonometric closed form for a start value (relative error ~1e-6 whatever the clustering);
//   3. fp64 Newton on det(A - x I) started just LEFT of the estimate: for a cubic with real roots the
//      iterates rise monotonically to the smallest root (quadratic when it is separated; when roots
//      cluster the start value is already inside the tolerance — or, if it landed right of them, the
//      iteration starts again from just left of zero);
//   4. eigenvector = largest cross product of two rows of (A - x I); rank-deficient fall-backs below.
// c = xx,xy,xz,yy,yz,zz.  Returns lambda (>= 0 up to rounding) and a unit vector.
GNDT_HD void min_eigenpair_sym3(const double c[6], double& lambda, double v[3]) {
    double s = fmax(fmax(fabs(c[0]), fabs(c[3])), fabs(c[5]));
    s = fmax(s, fmax(fmax(fabs(c[1]), fabs(c[2])), fabs(c[4])));
    v[0] = 0.0; v[1] = 0.0; v[2] = 1.0;
    lambda = 0.0;
    if (!(s > 0.0)) return;                       // all-zero scatter (identical points)
    const double inv = 1.0 / s;
    const double a00 = c[0] * inv, a01 = c[1] * inv, a02 = c[2] * inv, a11 = c[3] * inv, a12 = c[4] * inv, a22 = c[5] * inv;
    const double c2 = a00 + a11 + a22;
    const double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
    const double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    // fp32 start value
    const float q = (float)c2 * (1.0f / 3.0f);
    const float b00 = (float)a00 - q, b11 = (float)a11 - q, b22 = (float)a22 - q;
    const float f01 = (float)a01, f02 = (float)a02, f12 = (float)a12;
    const float p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0f * (f01 * f01 + f02 * f02 + f12 * f12);
    double lam;
    if (p2 < 1e-12f) {
        lam = (double)q - 1e-5;                    // (numerically) isotropic
    } else {
        const float p = sqrtf(p2 * (1.0f / 6.0f));
        const float ip = 1.0f / p;
        const float d00 = b00 * ip, d11 = b11 * ip, d22 = b22 * ip, d01 = f01 * ip, d02 = f02 * ip, d12 = f12 * ip;
        float r = 0.5f * (d00 * (d11 * d22 - d12 * d12) - d01 * (d01 * d22 - d12 * d02) + d02 * (d01 * d12 - d11 * d02));
        r = fminf(1.0f, fmaxf(-1.0f, r));
        const float phi = acosf(r) * (1.0f / 3.0f);
        lam = (double)(q + 2.0f * p * cosf(phi + 2.0943951023931953f)) - 1e-5;
    }
    // separated root: 2-4 iterations (quadratic); double/triple root: halves the error each time
    bool restarted = false;
    double prev = 1e300;
    for (int it = 0; it < 26; ++it) {
        const double f = ((c2 - lam) * lam - c1) * lam + c0;
        const double fp = (2.0 * c2 - 3.0 * lam) * lam - c1;
        if (!(fp < 0.0)) {
            // Beyond the cubic's first critical point.  On the way up from the left that means the roots coincide to rounding;
            // at the START it means the fp32 value itself was too high: with two clustered small roots (one scan line in a
            // cell: rank-1 scatter) acos() is taken near 1 and the estimate is up to ~1e-4 off, more than the 1e-5 it is
            // moved left by.  A scatter matrix has no root below zero: start again from just left of it (linear convergence
            // towards a double root: 26 halvings of 1e-5 .. 1e-4).
            if (it == 0 && !restarted) { restarted = true; lam = -1e-5; it = -1; continue; }
            break;
        }
        const double step = f / fp;
        // Left of the smallest of three real roots a Newton step is 1 / Sum 1 / (root_i - x): it shrinks from one iteration to
        // the next.  A step LARGER than the last one is rounding noise (f and f' both vanish at a double root: 3e-17 / 2e-12
        // threw the iterate 1e-5 to the right of a double root at zero) and is not taken.
        if (fabs(step) > prev) break;
        prev = fabs(step);
        lam -= step;
        if (fabs(step) <= 1e-15) break;
    }
    lambda = lam * s;
    // eigenvector
    const double m00 = a00 - lam, m11 = a11 - lam, m22 = a22 - lam;
    const double x0 = a01 * a12 - a02 * m11, y0 = a02 * a01 - m00 * a12, z0 = m00 * m11 - a01 * a01;   // r0 x r1
    const double x1 = a01 * m22 - a02 * a12, y1 = a02 * a02 - m00 * m22, z1 = m00 * a12 - a01 * a02;   // r0 x r2
    const double x2 = m11 * m22 - a12 * a12, y2 = a12 * a02 - a01 * m22, z2 = a01 * a12 - m11 * a02;   // r1 x r2
    const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
    if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
    // (the cross products carry ~1e-16 of absolute rounding: below |cross| = 1e-8 the direction they give is worse than what the
    //  rank-one fall-back gives — any vector orthogonal to the dominant row is then an eigenvector to ~1e-8)
    if (nn > 1e-16) {
        const double k = 1.0 / sqrt(nn);
        v[0] = vx * k; v[1] = vy * k; v[2] = vz * k;
        return;
    }
    // (A - x I) has rank <= 1: the eigenspace is a plane.  Take a unit vector orthogonal to the largest row.
    const double q0 = m00 * m00 + a01 * a01 + a02 * a02, q1 = a01 * a01 + m11 * m11 + a12 * a12, q2 = a02 * a02 + a12 * a12 + m22 * m22;
    double rx = m00, ry = a01, rz = a02, qq = q0;
    if (q1 > qq) { rx = a01; ry = m11; rz = a12; qq = q1; }
    if (q2 > qq) { rx = a02; ry = a12; rz = m22; qq = q2; }
    if (!(qq > 1e-26)) return;                     // A - x I == 0: every direction; keep (0,0,1)
    // r x e_k with k = the smallest component of r
    const double ax = fabs(rx), ay = fabs(ry), az = fabs(rz);
    if (ax <= ay && ax <= az) { vx = 0.0; vy = rz; vz = -ry; }
    else if (ay <= az) { vx = -rz; vy = 0.0; vz = rx; }
    else { vx = ry; vy = -rx; vz = 0.0; }
    const double k = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
    v[0] = vx * k; v[1] = vy * k; v[2] = vz * k;
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

// mean (fp32, as the reference stores it) and un-normalised scatter (fp64) from the additive statistics
GNDT_HD void node_moments(uint32_t n, const double sums[9], const double centre[3], float mean[3], double S[6]) {
    const double inv = 1.0 / (double)n;
    const double m[3] = {sums[0] * inv, sums[1] * inv, sums[2] * inv};
    S[0] = sums[3] - sums[0] * m[0];
    S[1] = sums[4] - sums[0] * m[1];
    S[2] = sums[5] - sums[0] * m[2];
    S[3] = sums[6] - sums[1] * m[1];
    S[4] = sums[7] - sums[1] * m[2];
    S[5] = sums[8] - sums[2] * m[2];
    // Noise floor.  S_kk = Sum v_k^2 - (Sum v_k)^2 / n is a difference of two fp64 sums of n terms, each
    // carrying up to ~n * 2^-53 relative rounding.  A diagonal at or below that floor means the node has no
    // extent along k at working precision (the reference's (0,0,0) padding: tens of thousands of identical
    // points): it is exactly zero, and so are its off-diagonals (|S_jk| <= sqrt(S_jj S_kk)).  This also
    // keeps the matrix positive semi-definite.
    const double floor_rel = 2.0 * (double)n * 1.1102230246251565e-16;
    const bool z0 = S[0] <= floor_rel * sums[3], z1 = S[3] <= floor_rel * sums[6], z2 = S[5] <= floor_rel * sums[8];
    if (z0) { S[0] = 0.0; S[1] = 0.0; S[2] = 0.0; }
    if (z1) { S[3] = 0.0; S[1] = 0.0; S[4] = 0.0; }
    if (z2) { S[5] = 0.0; S[2] = 0.0; S[4] = 0.0; }
    for (int k = 0; k < 3; ++k) mean[k] = (float)(centre[k] + m[k]);
}

// roughness and normal from the scatter (OcNode::countRoughNormal, map2D.h:110-133)
GNDT_HD void node_rough_normal(const double S[6], float& rough_out, float normal[3]) {
    double lam, v[3];
    min_eigenpair_sym3(S, lam, v);
    double nx = v[0], ny = v[1], nz = v[2];
    // sign is unspecified in the reference (consumers fold the angle, map2D.h:477-482): point it up
    if (nz < 0.0 || (nz == 0.0 && (ny < 0.0 || (ny == 0.0 && nx < 0.0)))) { nx = -nx; ny = -ny; nz = -nz; }
    float rough = (float)lam;
    if (rough < 0.f) rough = 0.f;
    if (rough == 0.f) rough = 0.01f;   // map2D.h:131-132
    rough_out = rough;
    normal[0] = (float)nx; normal[1] = (float)ny; normal[2] = (float)nz;
}

GNDT_HD void finalize_node(uint32_t n, const double sums[9], const double centre[3], NodeResult& r) {
    double S[6];
    node_moments(n, sums, centre, r.mean, S);
    for (int k = 0; k < 6; ++k) r.cov[k] = (float)S[k];
    node_rough_normal(S, r.rough, r.normal);
}

// mean z as the reference stores it (fp32), used by the slope test
GNDT_HD float node_mean_z(uint32_t n, double sum_vz, double centre_z) {
    return (float)(centre_z + sum_vz / (double)n);
}

// The zadd / zminus rule of OcNode::isSlope (include/map2D.h:69-75): there is no level 0.
GNDT_HD int level_above(int z) { return (z == -1) ? 1 : z + 1; }
GNDT_HD int level_below(int z) { return (z == 1) ? -1 : z - 1; }

}  // namespace gndt
