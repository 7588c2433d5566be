// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_cpu.cpp).  Key codec of the reference, shared by the oracle's
// translation units (ref_cpu.cpp: grid build; cost_cpu.cpp: cost map).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>

namespace {

// ------------------------------------------------------------------------------------------------
// Key codec (Stopwatch.h).  Written through strings on purpose: the wrap-around and the behaviour
// on negative values are properties of the string procedure.
// ------------------------------------------------------------------------------------------------

// Stopwatch.h:39-47 / 49-57: binary digits of n, most significant first; empty for 0.
// For negative n, C++ '%' yields -1 (truthy) and '/' truncates toward zero.
std::string binary_digits(int n) {
    std::string s;
    int a = n;
    while (a != 0) {
        s.push_back((a % 2) != 0 ? '1' : '0');
        a /= 2;
    }
    std::reverse(s.begin(), s.end());
    return s;
}

// Stopwatch.h:102-110: parse leading 0/1 characters into an UNSIGNED 32-bit accumulator (older bits
// fall off the top), then reinterpret as int.
int parse_binary_u32(const std::string& str) {
    uint32_t acc = 0;
    for (char c : str) {
        if (c != '0' && c != '1') break;
        acc = (acc << 1) | static_cast<uint32_t>(c - '0');
    }
    return static_cast<int>(acc);
}

// Stopwatch.h:116-147: pad the shorter digit string with leading zeros, interleave so that each
// pair is (digit of a, digit of b) from the most significant end, parse, print in decimal.
std::string count_morton(int a, int b) {
    std::string da = binary_digits(a), db = binary_digits(b);
    if (da.size() < db.size()) da.insert(0, db.size() - da.size(), '0');
    if (db.size() < da.size()) db.insert(0, da.size() - db.size(), '0');
    std::string inter;
    inter.reserve(2 * da.size());
    for (size_t i = 0; i < da.size(); ++i) {
        inter.push_back(da[i]);
        inter.push_back(db[i]);
    }
    return std::to_string(parse_binary_u32(inter));
}

// Stopwatch.h:171-189: inverse (valid for a <= 32767).
void morton_to_xy(int morton, int* a, int* b) {
    std::string m = binary_digits(morton);
    if (m.size() % 2 != 0) m.insert(0, 1, '0');
    std::string da, db;
    for (size_t i = 0; i + 1 < m.size(); i += 2) {
        da.push_back(m[i]);
        db.push_back(m[i + 1]);
    }
    *a = parse_binary_u32(da);
    *b = parse_binary_u32(db);
}

struct Key {
    char quadrant;  // 'A'..'D'  map2D.h:952-962
    int nx, ny;     // 1-based ceil indices  map2D.h:965-970
    int sz;         // signed z level (no level 0)  map2D.h:963-964, 973
};

// map2D.h:950-976.  All arithmetic in fp32; abs/ceil are the float overloads (SURVEY §8c probe).
inline Key trans_key(const float o[3], float gridLen, float zLen, float px, float py, float pz) {
    Key k;
    if (px > o[0]) k.quadrant = (py > o[1]) ? 'A' : 'B';
    else           k.quadrant = (py > o[1]) ? 'C' : 'D';
    int zsign = (pz > o[2]) ? 1 : -1;
    const float qx = std::fabs(px - o[0]) / gridLen;  // x86-64 SSE: every step rounds to fp32
    const float qy = std::fabs(py - o[1]) / gridLen;
    const float qz = std::fabs(pz - o[2]) / zLen;
    int nx = static_cast<int>(std::ceil(qx));
    int ny = static_cast<int>(std::ceil(qy));
    int nz = static_cast<int>(std::ceil(qz));
    if (nx == 0) nx = 1;
    if (ny == 0) ny = 1;
    if (nz == 0) nz = 1;
    k.nx = nx; k.ny = ny; k.sz = zsign * nz;
    return k;
}

inline int signed_x(const Key& k) { return (k.quadrant == 'A' || k.quadrant == 'B') ? k.nx : -k.nx; }
inline int signed_y(const Key& k) { return (k.quadrant == 'A' || k.quadrant == 'C') ? k.ny : -k.ny; }

}  // namespace
