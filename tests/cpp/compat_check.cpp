// Test program for include/gndt_compat.hpp (built and run by tests/test_compat_cpp.py).
//   compat_check <cloud.f32> <n> <grid_len> <z_len> <interval> <demand> [gpu]
// 1. runs the CPU oracle (liboracle.so: test infrastructure) on the cloud and materialises its export
//    into the reference-shaped containers; checks the container invariants the reference relies on;
// 2. with "gpu": builds the same cloud through gndt_compat::TwoDmap::create2DMap (libgndt, C ABI) and
//    compares the two container sets entry by entry (keys/order exact, values to 1e-5).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "gndt_compat.hpp"

extern "C" {
void* oracle_build(const float* xyz, size_t n, size_t stride_floats, const float origin[3], float grid_len, float z_len,
                   float slope_interval, int demand, int min_points, int mode, int threads);
size_t oracle_num_nodes(void* h);
size_t oracle_num_columns(void* h);
void oracle_export(void* h, int32_t* sx, int32_t* sy, int32_t* sz, uint32_t* count, uint64_t* first_idx, float* mean,
                   float* cov, float* evals, float* rough, float* normal, uint32_t* flags, double* mean64, double* cov64,
                   double* rough64, double* normal64, double* evals64, char* morton);
void oracle_free(void* h);
}

using namespace gndt_compat;

#define CHECK(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

// Container-level check: the exact 1e-5 gates (with the fp32 oracle's own rounding accounted for) live in
// tests/parity.py; here a looser 1e-4 guards against wiring mistakes (wrong row, transposed entries).
static bool close(float a, float b, float scale) { return std::fabs(a - b) <= 1e-4f * std::fmax(scale, 1e-30f); }

int main(int argc, char** argv) {
    if (argc < 7) { std::printf("usage\n"); return 2; }
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    const float gl = std::strtof(argv[3], nullptr), zl = std::strtof(argv[4], nullptr), iv = std::strtof(argv[5], nullptr);
    const std::string demand = argv[6];
    const bool gpu = argc > 7 && std::string(argv[7]) == "gpu";
    std::vector<float> cloud(3 * n);
    FILE* f = std::fopen(argv[1], "rb");
    CHECK(f && std::fread(cloud.data(), 4, 3 * n, f) == 3 * n, "cannot read cloud");
    std::fclose(f);

    // ---- oracle -> export -> materialise ----
    void* oh = oracle_build(cloud.data() + 3, n - 1, 3, cloud.data(), gl, zl, iv, demand == "true" ? 1 : 0, 3, 0, 1);
    const size_t C = oracle_num_nodes(oh), K = oracle_num_columns(oh);
    CellsHost ex;
    ex.resize(C);
    std::vector<uint64_t> first64(C);
    std::vector<char> morton(16 * C);
    oracle_export(oh, ex.sx.data(), ex.sy.data(), ex.sz.data(), ex.count.data(), first64.data(), ex.mean.data(), ex.cov.data(),
                  nullptr, ex.rough.data(), ex.normal.data(), ex.flags.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                  morton.data());
    oracle_free(oh);
    for (size_t i = 0; i < C; ++i) ex.first_idx[i] = (uint32_t)first64[i];
    ex.view.num_columns = K;
    TwoDmap A(gl, zl);
    materialise(ex.view, A);

    // invariants against the oracle's own strings and order
    CHECK(A.morton_list.size() == K, "morton_list %zu != %zu", A.morton_list.size(), K);
    CHECK(A.map_xy.size() == C, "map_xy %zu != %zu", A.map_xy.size(), C);
    CHECK(A.map_cell.size() == K, "map_cell %zu != %zu", A.map_cell.size(), K);
    {
        size_t i = 0, slopes = 0;
        std::set<std::string> seen;
        for (const std::string& key : A.morton_list) {
            CHECK(seen.insert(key).second, "duplicate column %s", key.c_str());
            CHECK(A.map_cell.count(key) == 1 && A.map_cell[key]->getMorton() == key, "cell %s", key.c_str());
            auto range = A.map_xy.equal_range(key);
            for (auto it = range.first; it != range.second; ++it, ++i) {
                CHECK(i < C, "too many nodes");
                CHECK(key == std::string(&morton[16 * i]), "row %zu: key %s != oracle %s", i, key.c_str(), &morton[16 * i]);
                const OcNode* nd = it->second;
                CHECK(nd->z == ex.sz[i] && nd->morton == key, "row %zu z/order", i);
                const bool has = ex.flags[i] & 1u;
                CHECK(nd->N == (has ? (int)ex.count[i] : 0), "row %zu N", i);
                CHECK(nd->xyz_centroid(2) == ex.mean[3 * i + 2], "row %zu centroid", i);
                CHECK(nd->covariance_matrix(1, 2) == ex.cov[6 * i + 4] && nd->covariance_matrix(2, 1) == ex.cov[6 * i + 4], "row %zu cov", i);
                const bool sl = ex.flags[i] & 2u;
                const Cell* cell = A.map_cell[key];
                CHECK((cell->map_slope.count(nd->z) == 1) == sl, "row %zu slope presence", i);
                if (sl) {
                    const Slope* s = cell->map_slope.at(nd->z);
                    CHECK(s->morton_xy == key && s->morton_z == nd->z && s->rough == ex.rough[i] && s->h == FLT_MAX && !s->up &&
                              s->down == ((ex.flags[i] & 4u) != 0) && s->father == nullptr && s->mean(0) == ex.mean[3 * i],
                          "row %zu slope fields", i);
                    ++slopes;
                }
            }
        }
        CHECK(i == C, "nodes visited %zu != %zu", i, C);
        // transMortonXYZ on a point must give the key of the node it was binned into
        A.setCloudFirst(Vector3f{{cloud[0], cloud[1], cloud[2]}});
        for (size_t p = 1; p < n; p += (n / 997) + 1) {
            Vector3f q{{cloud[3 * p], cloud[3 * p + 1], cloud[3 * p + 2]}};
            std::string key; int z;
            CHECK(A.transMortonXYZ(q, key, z), "transMortonXYZ range");
            bool found = false;
            auto range = A.map_xy.equal_range(key);
            for (auto it = range.first; it != range.second; ++it) found = found || it->second->z == z;
            CHECK(found, "point %zu maps to %s/%d which is not in the map", p, key.c_str(), z);
        }
        std::printf("oracle->materialise OK nodes=%zu columns=%zu slopes=%zu\n", C, K, slopes);
    }
    if (!gpu) return 0;

    // ---- libgndt through the reference-shaped wrapper ----
    TwoDmap B(gl, zl);
    B.setInterval(iv);
    B.setCloudFirst(Vector3f{{cloud[0], cloud[1], cloud[2]}});
    CHECK(B.create2DMap(demand, cloud.data() + 3, n - 1, 12), "create2DMap failed: %s", B.lastError().c_str());
    CHECK(B.morton_list == A.morton_list, "morton_list differs");
    CHECK(B.map_xy.size() == A.map_xy.size() && B.map_cell.size() == A.map_cell.size(), "sizes differ");
    auto ia = A.map_xy.begin();
    auto ib = B.map_xy.begin();
    for (; ia != A.map_xy.end(); ++ia, ++ib) {
        CHECK(ia->first == ib->first && ia->second->z == ib->second->z && ia->second->N == ib->second->N, "map_xy entry %s", ia->first.c_str());
        float sc = 0;
        for (int k = 0; k < 9; ++k) sc = std::fmax(sc, std::fabs(ia->second->covariance_matrix.m[k]));
        for (int k = 0; k < 9; ++k) CHECK(close(ia->second->covariance_matrix.m[k], ib->second->covariance_matrix.m[k], sc), "cov %s", ia->first.c_str());
        for (int k = 0; k < 3; ++k) CHECK(close(ia->second->xyz_centroid(k), ib->second->xyz_centroid(k), std::fmax(1.f, std::fabs(ia->second->xyz_centroid(k)))), "mean");
    }
    for (auto& kv : A.map_cell) {
        const Cell* cb = B.map_cell.at(kv.first);
        CHECK(cb->map_slope.size() == kv.second->map_slope.size(), "slopes of %s", kv.first.c_str());
        for (auto& s : kv.second->map_slope) {
            CHECK(cb->map_slope.count(s.first), "slope z");
            CHECK(cb->map_slope.at(s.first)->down == s.second->down, "down flag");
        }
    }
    std::printf("libgndt create2DMap == oracle OK\n");
    return 0;
}
