// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under grid_ndt_amd/ or include/ may include, link
// or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and only as the checker / reported CPU baseline.
//
// A CPU restatement of the reference's NDT grid-build path (daysun/grid_ndt):
//   src/receiver.cpp:41-93   uniformDivision      (find-or-create node, append point)
//   src/receiver.cpp:145-160 chatterCallback      (origin = point 0, bin 1..N-1, create2DMap)
//   include/map2D.h:950-976  transMortonXYZ       (quadrant + ceil-index key)
//   include/Stopwatch.h:39-47,59-64,102-110,116-147,171-189  Morton string helpers
//   include/map2D.h:592-668  create2DMap          (per-node Gaussian, slope label, eigen)
//   include/map2D.h:66-108   OcNode::isSlope      (order-dependent up/down test)
//   include/map2D.h:110-133  OcNode::countRoughNormal (min-eigen selection)
//
// PARITY STATUS: "parity unpinned" for the arithmetic.  The reference delegates mean/scatter to PCL
// `common` (compute3DCentroid / computeCovarianceMatrix, map2D.h:621-622) and the eigen-solve to
// Eigen (EigenSolver<Matrix3f>, map2D.h:111-113).  Neither library is vendored, version-pinned or
// installed here, and the reference has no tests.  Their published behaviour is restated:
//   * centroid  = sequential fp32 sum / n                       (PCL dense path)
//   * scatter   = sequential fp32 sum of (p-c)(p-c)^T, NOT divided by n
//   * eigen     = eigen-decomposition of the symmetric fp32 scatter; this file uses cyclic Jacobi in
//                 fp32 (Eigen's general real-Schur QR is not restated bit-for-bit), and picks the
//                 minimum with the reference's own comparison chain.
// The key codec IS pinned: tests/golden/morton_known_answers.csv holds the answers the compiled
// reference helpers gave in the survey container plus the header's own worked
// example (Stopwatch.h:112-115, 166-170).
//
// Three build modes share one export:
//   mode 0  "as shipped": single thread, decimal Morton strings, std::multimap<string,Node*>,
//           std::list<string> morton_list — the structure the reference actually executes.
//   mode 1  integer keys, single thread (same arithmetic, same order).
//   mode 2  integer keys, OpenMP over all host cores (the honest CPU baseline of BASELINE.md §3).
// Every mode also carries an fp64 "truth" (two-pass mean/scatter in double, Jacobi in double).
//
// Build: see oracle/Makefile  (g++ -O2 -fopenmp -ffp-contract=off; contraction must stay off so the
// fp32 sums round the way the reference's x86-64 build does).

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#include <parallel/algorithm>
#endif

#include "ref_codec.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// Node state (map2D.h:38-57) and result fields (Slope, map2D.h:136-146)
// ------------------------------------------------------------------------------------------------
struct Node {
    std::string morton;   // quadrant + decimal Morton (mode 0 only)
    Key key{};
    std::vector<uint32_t> idx;  // indices of this node's points in arrival order (stands for test_cloud)
    const uint32_t* idx_ptr = nullptr;  // mode 2 keeps them in one flat array per owner thread instead (same order)
    uint32_t idx_n = 0;
    uint64_t first_idx = 0;
    size_t count() const { return idx_ptr ? idx_n : idx.size(); }
    const uint32_t* points() const { return idx_ptr ? idx_ptr : idx.data(); }
    float mean[3] = {0, 0, 0};          // xyz_centroid, zero-initialised  map2D.h:55
    float cov[6] = {0, 0, 0, 0, 0, 0};  // upper triangle xx,xy,xz,yy,yz,zz of covariance_matrix  map2D.h:54
    int N = 0;                          // map2D.h:56
    bool slope = false, down = false;   // a Slope object exists / its 'down' flag
    float rough = 0.f, normal[3] = {0, 0, 0};
    float evals[3] = {0, 0, 0};
    // fp64 truth
    double mean64[3] = {0, 0, 0}, cov64[6] = {0, 0, 0, 0, 0, 0}, rough64 = 0, normal64[3] = {0, 0, 0};
    double evals64[3] = {0, 0, 0};
};

bool g_with_truth = true;    // the fp64 two-pass "truth" beside the reference arithmetic (oracle_set_truth)

struct Params {
    float grid_len, z_len, slope_interval;
    int demand;  // 0 = "slope", 1 = "true"
    int min_points;
};

// ---- Jacobi eigen-decomposition of a symmetric 3x3 (upper triangle in, eigenpairs out) ----
template <typename T>
void jacobi3(const T c[6], T evals[3], T evecs[3][3]) {
    T a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    T v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        T off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
        if (off == T(0)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == T(0)) continue;
                T theta = (a[q][q] - a[p][p]) / (T(2) * a[p][q]);
                T t = (theta >= T(0) ? T(1) : T(-1)) / (std::fabs(theta) + std::sqrt(theta * theta + T(1)));
                T cs = T(1) / std::sqrt(t * t + T(1)), sn = t * cs;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    T akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    T apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    T vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        evals[i] = a[i][i];
        for (int k = 0; k < 3; ++k) evecs[k][i] = v[k][i];
    }
}

// ---- general real eigen-solver of a 3x3 (what Eigen::EigenSolver<Matrix3f> runs, map2D.h:111-113) ----
// The reference does NOT call a symmetric solver: EigenSolver reduces the matrix to Hessenberg form with a Householder reflector,
// runs shifted (Francis) QR iterations to the real Schur form while accumulating the transformations, finds the eigenvectors of
// the triangular factor by back-substitution and transforms them back — the EISPACK orthes / hqr2 route (Wilkinson & Reinsch;
// published in JAMA, which Eigen's RealSchur and EigenSolver::doComputeEigenvectors cite as their source).  Restated here in the
// working precision T (fp32 for the reference's Matrix3f).  pseudoEigenvalueMatrix's diagonal = evals in the order the QR
// iteration deflates them (NOT sorted); pseudoEigenvectors' columns = evecs, NOT normalised (the consumers only use directions,
// map2D.h:477-482).  A 2x2 block whose discriminant rounds negative (two eigenvalues equal to working precision: Eigen would
// report a complex pair and the reference would read the block's diagonal) is given as a double eigenvalue with the Schur vectors.
// Eigen is un-vendored and un-versioned in the reference: this pins nothing — it bounds how much the choice of
// solver can matter (tests/test_oracle_eigensolver.py compares it with the Jacobi restatement on every fixed scene).
template <typename T>
void general_qr3(const T c[6], T evals[3], T evecs[3][3]) {
    constexpr int nn = 3;
    T H[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
    T V[3][3], ort[3] = {0, 0, 0}, d[3] = {0, 0, 0}, e[3] = {0, 0, 0};
    (void)e;                                            // (the imaginary parts: always zero on this path)
    const int low = 0, high = nn - 1;
    // orthes: Householder reduction to Hessenberg form
    for (int m = low + 1; m <= high - 1; ++m) {
        T scale = 0;
        for (int i = m; i <= high; ++i) scale += std::fabs(H[i][m - 1]);
        if (scale != T(0)) {
            T h = 0;
            for (int i = high; i >= m; --i) { ort[i] = H[i][m - 1] / scale; h += ort[i] * ort[i]; }
            T g = std::sqrt(h);
            if (ort[m] > 0) g = -g;
            h -= ort[m] * g;
            ort[m] -= g;
            for (int j = m; j < nn; ++j) {
                T f = 0;
                for (int i = high; i >= m; --i) f += ort[i] * H[i][j];
                f /= h;
                for (int i = m; i <= high; ++i) H[i][j] -= f * ort[i];
            }
            for (int i = 0; i <= high; ++i) {
                T f = 0;
                for (int j = high; j >= m; --j) f += ort[j] * H[i][j];
                f /= h;
                for (int j = m; j <= high; ++j) H[i][j] -= f * ort[j];
            }
            ort[m] = scale * ort[m];
            H[m][m - 1] = scale * g;
        }
    }
    for (int i = 0; i < nn; ++i) for (int j = 0; j < nn; ++j) V[i][j] = i == j ? T(1) : T(0);
    for (int m = high - 1; m >= low + 1; --m) {
        if (H[m][m - 1] != T(0)) {
            for (int i = m + 1; i <= high; ++i) ort[i] = H[i][m - 1];
            for (int j = m; j <= high; ++j) {
                T g = 0;
                for (int i = m; i <= high; ++i) g += ort[i] * V[i][j];
                g = (g / ort[m]) / H[m][m - 1];
                for (int i = m; i <= high; ++i) V[i][j] += g * ort[i];
            }
        }
    }
    // hqr2: Francis double-shift QR to the real Schur form
    int n = nn - 1;
    const T eps = std::numeric_limits<T>::epsilon();
    T exshift = 0, p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
    T norm = 0;
    for (int i = 0; i < nn; ++i)
        for (int j = std::max(i - 1, 0); j < nn; ++j) norm += std::fabs(H[i][j]);
    int iter = 0, total = 0;
    if (norm == T(0)) n = low - 1;                      // (the zero matrix is its own Schur form: Eigen's RealSchur skips the iteration too)
    while (n >= low && total < 40 * nn) {
        int l = n;
        while (l > low) {
            s = std::fabs(H[l - 1][l - 1]) + std::fabs(H[l][l]);
            if (s == T(0)) s = norm;
            if (std::fabs(H[l][l - 1]) <= eps * s) break;   // (<=, as Eigen's RealSchur tests it: an exact zero always deflates)
            --l;
        }
        if (l == n) {                                   // one root found
            H[n][n] += exshift;
            d[n] = H[n][n]; e[n] = 0;
            --n; iter = 0;
        } else if (l == n - 1) {                        // two roots found
            w = H[n][n - 1] * H[n - 1][n];
            p = (H[n - 1][n - 1] - H[n][n]) / T(2);
            q = p * p + w;
            z = std::sqrt(std::fabs(q));
            H[n][n] += exshift;
            H[n - 1][n - 1] += exshift;
            x = H[n][n];
            if (q >= 0) {                               // real pair
                z = p >= 0 ? p + z : p - z;
                d[n - 1] = x + z;
                d[n] = d[n - 1];
                if (z != T(0)) d[n] = x - w / z;
                e[n - 1] = 0; e[n] = 0;
                x = H[n][n - 1];
                s = std::fabs(x) + std::fabs(z);
                p = x / s; q = z / s;
                r = std::sqrt(p * p + q * q);
                p /= r; q /= r;
                for (int j = n - 1; j < nn; ++j) { z = H[n - 1][j]; H[n - 1][j] = q * z + p * H[n][j]; H[n][j] = q * H[n][j] - p * z; }
                for (int i = 0; i <= n; ++i) { z = H[i][n - 1]; H[i][n - 1] = q * z + p * H[i][n]; H[i][n] = q * H[i][n] - p * z; }
                for (int i = low; i <= high; ++i) { z = V[i][n - 1]; V[i][n - 1] = q * z + p * V[i][n]; V[i][n] = q * V[i][n] - p * z; }
            } else {                                    // discriminant rounded negative: a double eigenvalue to working precision
                d[n - 1] = x + p; d[n] = x + p;
                e[n - 1] = 0; e[n] = 0;
                H[n][n - 1] = 0;                        // (the block is taken as diagonal: its Schur vectors stand for the eigenvectors)
                H[n - 1][n - 1] = d[n - 1]; H[n][n] = d[n];
            }
            n -= 2; iter = 0;
        } else {                                        // no convergence yet
            x = H[n][n]; y = 0; w = 0;
            if (l < n) { y = H[n - 1][n - 1]; w = H[n][n - 1] * H[n - 1][n]; }
            if (iter == 10) {                           // Wilkinson's original ad hoc shift
                exshift += x;
                for (int i = low; i <= n; ++i) H[i][i] -= x;
                s = std::fabs(H[n][n - 1]) + std::fabs(H[n - 1][n - 2]);
                x = y = T(0.75) * s;
                w = T(-0.4375) * s * s;
            }
            if (iter == 30) {                           // MATLAB's ad hoc shift
                s = (y - x) / T(2);
                s = s * s + w;
                if (s > 0) {
                    s = std::sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / T(2) + s);
                    for (int i = low; i <= n; ++i) H[i][i] -= s;
                    exshift += s;
                    x = y = w = T(0.964);
                }
            }
            ++iter; ++total;
            int m = n - 2;
            while (m >= l) {
                z = H[m][m];
                r = x - z; s = y - z;
                p = (r * s - w) / H[m + 1][m] + H[m][m + 1];
                q = H[m + 1][m + 1] - z - r - s;
                r = H[m + 2][m + 1];
                s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                p /= s; q /= s; r /= s;
                if (m == l) break;
                if (std::fabs(H[m][m - 1]) * (std::fabs(q) + std::fabs(r)) <
                    eps * (std::fabs(p) * (std::fabs(H[m - 1][m - 1]) + std::fabs(z) + std::fabs(H[m + 1][m + 1])))) break;
                --m;
            }
            for (int i = m + 2; i <= n; ++i) { H[i][i - 2] = 0; if (i > m + 2) H[i][i - 3] = 0; }
            for (int k = m; k <= n - 1; ++k) {          // double QR step on rows l:n, columns m:n
                const bool notlast = k != n - 1;
                if (k != m) {
                    p = H[k][k - 1]; q = H[k + 1][k - 1];
                    r = notlast ? H[k + 2][k - 1] : T(0);
                    x = std::fabs(p) + std::fabs(q) + std::fabs(r);
                    if (x == T(0)) continue;
                    p /= x; q /= x; r /= x;
                }
                s = std::sqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != T(0)) {
                    if (k != m) H[k][k - 1] = -s * x;
                    else if (l != m) H[k][k - 1] = -H[k][k - 1];
                    p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                    for (int j = k; j < nn; ++j) {
                        p = H[k][j] + q * H[k + 1][j];
                        if (notlast) { p += r * H[k + 2][j]; H[k + 2][j] -= p * z; }
                        H[k][j] -= p * x;
                        H[k + 1][j] -= p * y;
                    }
                    for (int i = 0; i <= std::min(n, k + 3); ++i) {
                        p = x * H[i][k] + y * H[i][k + 1];
                        if (notlast) { p += z * H[i][k + 2]; H[i][k + 2] -= p * r; }
                        H[i][k] -= p;
                        H[i][k + 1] -= p * q;
                    }
                    for (int i = low; i <= high; ++i) {
                        p = x * V[i][k] + y * V[i][k + 1];
                        if (notlast) { p += z * V[i][k + 2]; V[i][k + 2] -= p * r; }
                        V[i][k] -= p;
                        V[i][k + 1] -= p * q;
                    }
                }
            }
        }
    }
    if (norm == T(0)) { for (int i = 0; i < nn; ++i) d[i] = 0; }
    while (n >= low) { d[n] = H[n][n] + exshift; e[n] = 0; --n; }      // (iteration limit: what is on the diagonal)
    // back-substitution: eigenvectors of the (quasi-)triangular factor, all roots real here
    if (norm != T(0)) {
        for (n = nn - 1; n >= 0; --n) {
            p = d[n];
            H[n][n] = 1;
            for (int i = n - 1; i >= 0; --i) {
                w = H[i][i] - p;
                r = 0;
                for (int j = i + 1; j <= n; ++j) r += H[i][j] * H[j][n];
                H[i][n] = w != T(0) ? -r / w : -r / (eps * norm);
                t = std::fabs(H[i][n]);
                if ((eps * t) * t > 1) for (int j = i; j <= n; ++j) H[j][n] /= t;     // overflow control
            }
        }
        for (int j = nn - 1; j >= low; --j)             // back transformation to the eigenvectors of the original matrix
            for (int i = low; i <= high; ++i) {
                z = 0;
                for (int k = low; k <= std::min(j, high); ++k) z += V[i][k] * H[k][j];
                V[i][j] = z;
            }
    }
    for (int i = 0; i < 3; ++i) {
        evals[i] = d[i];
        for (int k = 0; k < 3; ++k) evecs[k][i] = V[k][i];
    }
}

// map2D.h:114-130: index of the chosen eigenvalue; ties go to the higher index.
template <typename T>
inline int pick_min(const T e[3]) {
    if (e[0] < e[1]) return (e[0] < e[2]) ? 0 : 2;
    return (e[1] < e[2]) ? 1 : 2;
}

// map2D.h:611-627 (+ PCL semantics) and :110-133 for one node.
void fit_node(Node& nd, const float* xyz, size_t stride, const Params& P) {
    const size_t n = nd.count();
    const uint32_t* const pts = nd.points();
    if (n < static_cast<size_t>(P.min_points)) return;  // map2D.h:611; mean/cov/N stay zero
    // pcl::compute3DCentroid, dense path: sequential fp32 sums, then divide.
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (size_t q_ = 0; q_ < n; ++q_) {
        const uint32_t i = pts[q_];
        const float* p = xyz + static_cast<size_t>(i) * stride;
        sx += p[0]; sy += p[1]; sz += p[2];
    }
    const float fn = static_cast<float>(n);
    nd.mean[0] = sx / fn; nd.mean[1] = sy / fn; nd.mean[2] = sz / fn;
    // pcl::computeCovarianceMatrix(cloud, centroid, C): sequential fp32, un-normalised.
    float c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (size_t q_ = 0; q_ < n; ++q_) {
        const uint32_t i = pts[q_];
        const float* p = xyz + static_cast<size_t>(i) * stride;
        float dx = p[0] - nd.mean[0], dy = p[1] - nd.mean[1], dz = p[2] - nd.mean[2];
        c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
        c00 += dx * dx; c01 += dx * dy; c02 += dx * dz;
    }
    nd.cov[0] = c00; nd.cov[1] = c01; nd.cov[2] = c02; nd.cov[3] = c11; nd.cov[4] = c12; nd.cov[5] = c22;
    nd.N += static_cast<int>(n);  // map2D.h:625

    if (!g_with_truth) return;      // (the timed CPU baseline does only what the reference does)
    // fp64 truth: two-pass in double
    double m[3] = {0, 0, 0};
    for (size_t q_ = 0; q_ < n; ++q_) {
        const uint32_t i = pts[q_];
        const float* p = xyz + static_cast<size_t>(i) * stride;
        m[0] += p[0]; m[1] += p[1]; m[2] += p[2];
    }
    for (int k = 0; k < 3; ++k) { m[k] /= static_cast<double>(n); nd.mean64[k] = m[k]; }
    double d[6] = {0, 0, 0, 0, 0, 0};
    for (size_t q_ = 0; q_ < n; ++q_) {
        const uint32_t i = pts[q_];
        const float* p = xyz + static_cast<size_t>(i) * stride;
        double dx = p[0] - m[0], dy = p[1] - m[1], dz = p[2] - m[2];
        d[0] += dx * dx; d[1] += dx * dy; d[2] += dx * dz; d[3] += dy * dy; d[4] += dy * dz; d[5] += dz * dz;
    }
    for (int k = 0; k < 6; ++k) nd.cov64[k] = d[k];
}

int g_eigen_solver = 0;      // 0: cyclic Jacobi (symmetric); 1: Hessenberg + shifted QR + back-substitution, EigenSolver's route
void eigen_node(Node& nd) {  // OcNode::countRoughNormal, map2D.h:110-133
    float ev[3], vec[3][3];
    if (g_eigen_solver == 1) general_qr3<float>(nd.cov, ev, vec);
    else jacobi3<float>(nd.cov, ev, vec);
    int j = pick_min(ev);
    nd.rough = ev[j];
    for (int k = 0; k < 3; ++k) { nd.normal[k] = vec[k][j]; nd.evals[k] = ev[k]; }
    if (nd.rough == 0.f) nd.rough = 0.01f;  // map2D.h:131-132
    if (!g_with_truth) return;
    double ev64[3], vec64[3][3];
    jacobi3<double>(nd.cov64, ev64, vec64);
    int j64 = pick_min(ev64);
    nd.rough64 = ev64[j64];
    for (int k = 0; k < 3; ++k) { nd.normal64[k] = vec64[k][j64]; nd.evals64[k] = ev64[k]; }
}

// OcNode::isSlope, map2D.h:66-108, evaluated against the column's nodes in insertion order, reading
// the centroids as they are at call time.
bool is_slope(const Node& me, const std::vector<Node*>& column, float interval, bool& up, bool& down, int min_points) {
    if (me.N < min_points) return false;
    int zadd = me.key.sz + 1, zminus = me.key.sz - 1;
    if (me.key.sz == -1) zadd = 1;
    else if (me.key.sz == 1) zminus = -1;
    bool zup = false, zdown = false;
    for (const Node* o : column) {
        if (zup && zdown) break;
        if (zminus == o->key.sz && std::fabs(o->mean[2] - me.mean[2]) > interval) { zdown = true; down = true; }
        if (zadd == o->key.sz && std::fabs(o->mean[2] - me.mean[2]) > interval) { zup = true; up = true; }
    }
    return !up;
}

// create2DMap body for one column (map2D.h:606-662)
void process_column(std::vector<Node*>& column, const float* xyz, size_t stride, const Params& P) {
    for (Node* nd : column) {
        if (nd->count() < static_cast<size_t>(P.min_points)) continue;
        fit_node(*nd, xyz, stride, P);
        bool up = false, down = false;
        if (P.demand == 0) {
            if (is_slope(*nd, column, P.slope_interval, up, down, P.min_points)) {
                nd->slope = true; nd->down = down;
                eigen_node(*nd);
            }
        } else {
            nd->slope = true; nd->down = false;
            eigen_node(*nd);
        }
    }
}

struct Grid {
    std::vector<Node*> nodes;        // export order: first-seen column, then first-seen node
    std::vector<uint32_t> col_start; // index into nodes of each column's first node (+ sentinel)
    double t_division = 0, t_calculate = 0;
    // mode 2: nodes and their point lists live in per-thread pools
    std::vector<std::vector<Node>> node_pool;
    std::vector<std::vector<uint32_t>> idx_pool;
    ~Grid() { if (node_pool.empty()) for (Node* n : nodes) delete n; }
};

double now_s() {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

inline uint64_t pack_key(int sx, int sy, int sz) {
    return (static_cast<uint64_t>(static_cast<uint32_t>(sx + (1 << 20)) & 0x1FFFFF) << 43) |
           (static_cast<uint64_t>(static_cast<uint32_t>(sy + (1 << 20)) & 0x1FFFFF) << 22) |
           (static_cast<uint64_t>(static_cast<uint32_t>(sz + (1 << 21)) & 0x3FFFFF));
}

// ---- mode 0: the structure the reference executes -----------------------------------------------
Grid* build_as_shipped(const float* xyz, size_t n, size_t stride, const float o[3], const Params& P) {
    Grid* g = new Grid;
    std::multimap<std::string, Node*> map_xy;   // map2D.h:485
    std::list<std::string> morton_list;         // map2D.h:504
    double t0 = now_s();
    for (size_t i = 0; i < n; ++i) {            // receiver.cpp:150-154
        const float* p = xyz + i * stride;
        Key k = trans_key(o, P.grid_len, P.z_len, p[0], p[1], p[2]);
        std::string morton_xy(1, k.quadrant);
        morton_xy += count_morton(k.nx, k.ny);  // map2D.h:971-972
        if (map_xy.count(morton_xy) == 0) {     // receiver.cpp:60-70
            Node* nd = new Node; nd->morton = morton_xy; nd->key = k; nd->first_idx = i;
            nd->idx.push_back(static_cast<uint32_t>(i));
            map_xy.insert({morton_xy, nd});
            morton_list.push_back(morton_xy);
        } else {                                // receiver.cpp:71-91
            auto range = map_xy.equal_range(morton_xy);
            bool found = false;
            for (auto it = range.first; it != range.second; ++it)
                if (it->second->key.sz == k.sz) { it->second->idx.push_back(static_cast<uint32_t>(i)); found = true; break; }
            if (!found) {
                Node* nd = new Node; nd->morton = morton_xy; nd->key = k; nd->first_idx = i;
                nd->idx.push_back(static_cast<uint32_t>(i));
                map_xy.insert({morton_xy, nd});  // equal keys keep insertion order (C++11)
            }
        }
    }
    double t1 = now_s();
    for (const std::string& key : morton_list) {  // map2D.h:595-664
        auto range = map_xy.equal_range(key);
        std::vector<Node*> column;
        for (auto it = range.first; it != range.second; ++it) column.push_back(it->second);
        process_column(column, xyz, stride, P);
        g->col_start.push_back(static_cast<uint32_t>(g->nodes.size()));
        for (Node* nd : column) g->nodes.push_back(nd);
    }
    g->col_start.push_back(static_cast<uint32_t>(g->nodes.size()));
    double t2 = now_s();
    g->t_division = t1 - t0; g->t_calculate = t2 - t1;
    return g;
}

// ---- modes 1/2: integer keys; mode 2 shards nodes over OpenMP threads by key hash -----------------
inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// Mode 1 (threads == 1) and mode 2 (OpenMP): the CPU baseline bench.py times.  Same arithmetic as mode 0 — every node
// sees its points in arrival order, which the reference's sequential fp32 sums depend on — but none of its containers:
//   keys      packed integer key of every point                                             parallel over points
//   partition counting sort of the point indices by OWNER = hash(column) mod threads, stable  two parallel passes, flat
//   group     every owner: open-addressing table key -> node, then the nodes' point lists as  parallel over owners,
//             one flat array (count, prefix, fill): no per-node allocation, no vector growth    no sharing
//   order     columns by first-seen index (all nodes of a column sit with one owner), nodes of  parallel sort
//             a column by first-seen index
//   calculate create2DMap's per-column body (process_column)                                 parallel over columns
Grid* build_int_keys(const float* xyz, size_t n, size_t stride, const float o[3], const Params& P, int threads) {
    Grid* g = new Grid;
    if (threads < 1) threads = 1;
    const size_t T = static_cast<size_t>(threads);
    double t0 = now_s();
    // (Arrays of n elements are allocated WITHOUT a value-initialising pass: a std::vector would be zero-filled by this one
    // thread — 340 MB at 10 M points, a third of the "division" time on a 128-core box — and every page would then sit on
    // this thread's memory node.  Left uninitialised, a page is first touched by the thread of the loop that fills it.)
    std::unique_ptr<uint64_t[]> keys(new uint64_t[n]);
    std::unique_ptr<Key[]> kk(new Key[n]);
    std::unique_ptr<uint16_t[]> owner(new uint16_t[n]);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long i = 0; i < static_cast<long long>(n); ++i) {
        const float* p = xyz + static_cast<size_t>(i) * stride;
        Key k = trans_key(o, P.grid_len, P.z_len, p[0], p[1], p[2]);
        kk[i] = k;
        keys[i] = pack_key(signed_x(k), signed_y(k), k.sz);
        owner[i] = static_cast<uint16_t>(mix64(keys[i] & ~0x3FFFFFull) % T);      // by COLUMN: a column's nodes share an owner
    }
    // stable counting sort of the indices by owner: cnt[chunk][owner] -> offsets -> perm
    std::vector<size_t> cnt(T * T, 0), start(T + 1, 0);
    std::unique_ptr<uint32_t[]> perm(new uint32_t[n]);
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const size_t t = static_cast<size_t>(omp_get_thread_num());
#else
        const size_t t = 0;
#endif
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        size_t* mine = &cnt[t * T];
        for (size_t i = lo; i < hi; ++i) ++mine[owner[i]];
#pragma omp barrier
#pragma omp single
        {
            size_t run = 0;
            for (size_t ow = 0; ow < T; ++ow) {
                start[ow] = run;
                for (size_t c = 0; c < T; ++c) { const size_t v = cnt[c * T + ow]; cnt[c * T + ow] = run; run += v; }
            }
            start[T] = run;
        }
        for (size_t i = lo; i < hi; ++i) perm[mine[owner[i]]++] = static_cast<uint32_t>(i);
    }
    // every owner groups its points into nodes
    g->node_pool.resize(T);
    g->idx_pool.resize(T);
    struct Col { uint64_t first; uint32_t owner, begin, end; };      // a column = nodes [begin, end) of its owner's pool
    std::vector<std::vector<Col>> cols_of(T);
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        const size_t t = static_cast<size_t>(omp_get_thread_num());
#else
        const size_t t = 0;
#endif
        const size_t lo = start[t], hi = start[t + 1], m = hi - lo;
        size_t cap = 16;
        while (cap < 2 * m) cap <<= 1;
        std::vector<uint64_t> tkey(cap, ~0ull);
        std::vector<uint32_t> tval(cap), nid(m), ncount;
        std::vector<Node>& pool = g->node_pool[t];
        pool.reserve(m / 4 + 16);
        for (size_t q = 0; q < m; ++q) {
            const uint32_t i = perm[lo + q];
            const uint64_t k = keys[i];
            size_t s = static_cast<size_t>(mix64(k) >> 20) & (cap - 1);
            while (tkey[s] != ~0ull && tkey[s] != k) s = (s + 1) & (cap - 1);
            if (tkey[s] == ~0ull) {
                tkey[s] = k; tval[s] = static_cast<uint32_t>(pool.size());
                pool.emplace_back();
                pool.back().key = kk[i]; pool.back().first_idx = i;
                ncount.push_back(0);
            }
            nid[q] = tval[s];
            ++ncount[tval[s]];
        }
        std::vector<uint32_t>& flat = g->idx_pool[t];
        flat.resize(m);
        std::vector<uint32_t> fill(pool.size());
        uint32_t run = 0;
        for (size_t v = 0; v < pool.size(); ++v) { fill[v] = run; pool[v].idx_ptr = flat.data() + run; pool[v].idx_n = ncount[v]; run += ncount[v]; }
        for (size_t q = 0; q < m; ++q) flat[fill[nid[q]]++] = perm[lo + q];      // arrival order kept inside every node
        // columns of this owner: sort its nodes by (column key, first-seen); a column's first-seen = its first node's
        std::vector<uint32_t> ord(pool.size());
        for (size_t v = 0; v < ord.size(); ++v) ord[v] = static_cast<uint32_t>(v);
        auto colkey = [&](uint32_t v) { return pack_key(signed_x(pool[v].key), signed_y(pool[v].key), 0); };
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
            const uint64_t ca = colkey(a), cb = colkey(b);
            return ca != cb ? ca < cb : pool[a].first_idx < pool[b].first_idx;
        });
        std::vector<Node> sorted;
        sorted.reserve(pool.size());
        for (uint32_t v : ord) sorted.push_back(pool[v]);
        pool.swap(sorted);
        for (size_t v = 0; v < pool.size();) {
            size_t e = v + 1;
            const uint64_t ck = colkey(static_cast<uint32_t>(v));
            while (e < pool.size() && colkey(static_cast<uint32_t>(e)) == ck) ++e;
            cols_of[t].push_back(Col{pool[v].first_idx, static_cast<uint32_t>(t), static_cast<uint32_t>(v), static_cast<uint32_t>(e)});
            v = e;
        }
    }
    double t1 = now_s();
    // global column order = first-seen order (morton_list, receiver.cpp:70)
    std::vector<Col> cols;
    for (auto& v : cols_of) cols.insert(cols.end(), v.begin(), v.end());
#ifdef _OPENMP
    __gnu_parallel::sort(cols.begin(), cols.end(), [](const Col& a, const Col& b) { return a.first < b.first; }, __gnu_parallel::default_parallel_tag(threads));
#else
    std::sort(cols.begin(), cols.end(), [](const Col& a, const Col& b) { return a.first < b.first; });
#endif
    g->col_start.resize(cols.size() + 1);
    size_t total = 0;
    for (size_t c = 0; c < cols.size(); ++c) { g->col_start[c] = static_cast<uint32_t>(total); total += cols[c].end - cols[c].begin; }
    g->col_start[cols.size()] = static_cast<uint32_t>(total);
    g->nodes.resize(total);
    const long long ncol = static_cast<long long>(cols.size());
#pragma omp parallel for num_threads(threads) schedule(dynamic, 256)
    for (long long c = 0; c < ncol; ++c) {
        const Col& cl = cols[c];
        std::vector<Node*> column;
        for (uint32_t v = cl.begin; v < cl.end; ++v) column.push_back(&g->node_pool[cl.owner][v]);
        process_column(column, xyz, stride, P);
        std::copy(column.begin(), column.end(), g->nodes.begin() + g->col_start[c]);
    }
    double t2 = now_s();
    g->t_division = t1 - t0; g->t_calculate = t2 - t1;
    // mode 0 keeps the key string; fill it here too so that the exports agree.  Not part of the timed work: the
    // integer-key modes never need the string.
    const long long nn = static_cast<long long>(g->nodes.size());
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long i = 0; i < nn; ++i) {
        Node* nd = g->nodes[i];
        nd->morton = std::string(1, nd->key.quadrant) + count_morton(nd->key.nx, nd->key.ny);
    }
    return g;
}

}  // namespace

extern "C" {

void oracle_set_truth(int on) { g_with_truth = on != 0; }
void oracle_set_eigen_solver(int which) { g_eigen_solver = which == 1 ? 1 : 0; }

// ---- key codec entry points ---------------------------------------------------------------------
int oracle_count_morton(int a, int b, char* out, int cap) {
    std::string s = count_morton(a, b);
    if (static_cast<int>(s.size()) + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return 0;
}
void oracle_morton_to_xy(int morton, int* a, int* b) { morton_to_xy(morton, a, b); }

// transMortonXYZ: out_key receives quadrant letter + decimal Morton, NUL-terminated (cap >= 16)
void oracle_trans_morton_xyz(const float origin[3], float grid_len, float z_len, const float p[3],
                             char* out_key, int* nx, int* ny, int* sz) {
    Key k = trans_key(origin, grid_len, z_len, p[0], p[1], p[2]);
    std::string s(1, k.quadrant);
    s += count_morton(k.nx, k.ny);
    std::memcpy(out_key, s.c_str(), s.size() + 1);
    *nx = k.nx; *ny = k.ny; *sz = k.sz;
}

// ---- grid build ---------------------------------------------------------------------------------
// xyz: n points, stride_floats floats apart (3 = packed, 4 = pcl::PointXYZ).  The caller passes the
// cloud WITHOUT point 0 and gives point 0 as origin (receiver.cpp:145, 150).
void* oracle_build(const float* xyz, size_t n, size_t stride_floats, const float origin[3],
                   float grid_len, float z_len, float slope_interval, int demand, int min_points,
                   int mode, int threads) {
    Params P{grid_len, z_len, slope_interval, demand, min_points};
    if (threads <= 0) {
#ifdef _OPENMP
        threads = omp_get_max_threads();
#else
        threads = 1;
#endif
    }
    if (mode == 0) return build_as_shipped(xyz, n, stride_floats, origin, P);
    return build_int_keys(xyz, n, stride_floats, origin, P, mode == 1 ? 1 : threads);
}

size_t oracle_num_nodes(void* h) { return static_cast<Grid*>(h)->nodes.size(); }
size_t oracle_num_columns(void* h) { return static_cast<Grid*>(h)->col_start.size() - 1; }
void oracle_times(void* h, double* division_s, double* calculate_s) {
    Grid* g = static_cast<Grid*>(h);
    *division_s = g->t_division; *calculate_s = g->t_calculate;
}
int oracle_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// SoA export, nodes in reference order.  Any pointer may be NULL.
// flags: bit0 has_stats (N>=min_points), bit1 slope object exists, bit2 slope.down
void oracle_export(void* h, int32_t* sx, int32_t* sy, int32_t* sz, uint32_t* count, uint64_t* first_idx,
                   float* mean /*[n][3]*/, float* cov /*[n][6]*/, float* evals /*[n][3]*/, float* rough,
                   float* normal /*[n][3]*/, uint32_t* flags, double* mean64, double* cov64,
                   double* rough64, double* normal64, double* evals64, char* morton /*[n][16]*/) {
    Grid* g = static_cast<Grid*>(h);
    for (size_t i = 0; i < g->nodes.size(); ++i) {
        const Node& nd = *g->nodes[i];
        if (sx) sx[i] = signed_x(nd.key);
        if (sy) sy[i] = signed_y(nd.key);
        if (sz) sz[i] = nd.key.sz;
        if (count) count[i] = static_cast<uint32_t>(nd.count());
        if (first_idx) first_idx[i] = nd.first_idx;
        for (int k = 0; k < 3; ++k) {
            if (mean) mean[3 * i + k] = nd.mean[k];
            if (normal) normal[3 * i + k] = nd.normal[k];
            if (evals) evals[3 * i + k] = nd.evals[k];
            if (mean64) mean64[3 * i + k] = nd.mean64[k];
            if (normal64) normal64[3 * i + k] = nd.normal64[k];
            if (evals64) evals64[3 * i + k] = nd.evals64[k];
        }
        for (int k = 0; k < 6; ++k) {
            if (cov) cov[6 * i + k] = nd.cov[k];
            if (cov64) cov64[6 * i + k] = nd.cov64[k];
        }
        if (rough) rough[i] = nd.rough;
        if (rough64) rough64[i] = nd.rough64;
        if (flags) flags[i] = (nd.N > 0 ? 1u : 0u) | (nd.slope ? 2u : 0u) | (nd.down ? 4u : 0u);
        if (morton) {
            std::memset(morton + 16 * i, 0, 16);
            std::memcpy(morton + 16 * i, nd.morton.c_str(), std::min<size_t>(15, nd.morton.size()));
        }
    }
}

void oracle_free(void* h) { delete static_cast<Grid*>(h); }

}  // extern "C"
