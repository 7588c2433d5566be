// gndt_compat.hpp — dependency-free C++ mirror of the reference's map data model for the grid-build
// path, filled from libgndt's SoA export.  Header-only; needs only include/gndt.h + libgndt.so.
//
// The reference keeps its map in a global `daysun::TwoDmap map2D` (src/receiver.cpp:35) whose public
// containers are read by computeCost / CollisionCheck / AccessibleNeighbors (include/map2D.h:1285-1397,
// 351-475, 530-590) and by AstarPlanar (include/GlobalPlan.h:49-166):
//     multimap<string,OcNode*> map_xy      include/map2D.h:485
//     list<string>             morton_list include/map2D.h:504
//     map<string,Cell*>        map_cell    include/map2D.h:507
//     Cell::map_slope : map<int,Slope*>    include/map2D.h:184
// This header re-declares those shapes with the same member names (Eigen / octomath vector types are
// replaced by a 3-float struct with the same operator()(i) accessor) and rebuilds them with identical
// keys, iteration order and field values from a gndt_cells export.  A ROS/PCL/Eigen build of the
// reference would instead include its own map2D.h and use gndt_compat::materialise_into() as a template
// over its types.
#pragma once
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "gndt.h"

namespace gndt_compat {

struct Vector3f {                       // stands for Eigen::Vector3f and octomath::Vector3
    float d[3] = {0.f, 0.f, 0.f};
    float& operator()(int i) { return d[i]; }
    const float& operator()(int i) const { return d[i]; }
};
struct Matrix3f {                       // stands for Eigen::Matrix3f
    float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float& operator()(int r, int c) { return m[3 * r + c]; }
    const float& operator()(int r, int c) const { return m[3 * r + c]; }
};

// include/map2D.h:38-57.  test_cloud / lPoints are not mirrored: create2DMap drops the points
// (map2D.h:627); nodes below MINPOINTSIZE keep N == 0 and zero statistics like the reference's.
struct OcNode {
    Matrix3f covariance_matrix;
    Vector3f xyz_centroid;
    int N = 0;
    std::string morton;
    bool _isSlope = false;
    int z = 0;
    uint32_t count = 0;                 // extra: points binned (the reference's test_cloud.size() before the drop)
    bool isEmpty() const { return N < 3; }   // map2D.h:59-63
};

// include/map2D.h:136-146
struct Slope {
    Vector3f normal;
    float rough = 0.f;
    Vector3f mean;
    float h = FLT_MAX, g = FLT_MAX, f = FLT_MAX;     // map2D.h:637
    std::string morton_xy;
    int morton_z = 0;
    bool up = false, down = false;                   // `up` is not assigned by create2DMap (map2D.h:636)
    Slope* father = nullptr;
};

// include/map2D.h:181-187
class Cell {
    std::string morton;
public:
    std::map<int, Slope*> map_slope;
    explicit Cell(const std::string& m) : morton(m) {}
    std::string getMorton() const { return morton; }
};

// Host copy of a gndt_cells export (owning).
struct CellsHost {
    std::vector<int32_t> sx, sy, sz;
    std::vector<uint32_t> count, first_idx, flags;
    std::vector<float> mean, cov, rough, normal;
    gndt_cells view{};
    void resize(size_t n) {
        sx.resize(n); sy.resize(n); sz.resize(n); count.resize(n); first_idx.resize(n); flags.resize(n);
        mean.resize(3 * n); cov.resize(6 * n); rough.resize(n); normal.resize(3 * n);
        view.num_nodes = n;
        view.sx = sx.data(); view.sy = sy.data(); view.sz = sz.data(); view.count = count.data();
        view.first_idx = first_idx.data(); view.mean = mean.data(); view.cov = cov.data(); view.rough = rough.data();
        view.normal = normal.data(); view.flags = flags.data();
    }
};

// The map key of a column: quadrant letter + decimal Morton string, exactly what transMortonXYZ builds
// (map2D.h:952-972): A (+,+)  B (+,-)  C (-,+)  D (-,-).
inline std::string column_key(int32_t sx, int32_t sy) {
    char buf[16];
    gndt_count_morton(sx < 0 ? -sx : sx, sy < 0 ? -sy : sy, buf);
    const char q = (sx > 0) ? ((sy > 0) ? 'A' : 'B') : ((sy > 0) ? 'C' : 'D');
    return std::string(1, q) + buf;
}

// Where the element objects of one materialised map live: three allocations instead of three per node (round 3: 1.1-3.7 us per
// node went into `new` + tree inserts — 96 ms for an 80 k-node frame behind a 0.06 ms build).  The reference allocates every
// object with `new` and never frees it (src/receiver.cpp:62,85; map2D.h:598,632,648); a map filled through an arena is freed
// with the arena instead, and row_slope[i] is the Slope of result row i (nullptr: not a slope) for whoever writes per-row
// values back (apply_cost_into).
template <class Node, class SlopeT, class CellT>
struct MapArena {
    std::vector<Node> nodes;
    std::vector<SlopeT> slopes;
    std::vector<CellT> cells;
    std::vector<SlopeT*> row_slope;
    void clear() { nodes.clear(); slopes.clear(); cells.clear(); row_slope.clear(); }
};

namespace detail {
struct ColumnRun { uint64_t k_hi, k_lo; uint32_t first, count; };      // key characters packed big-endian: numeric order == std::string order
inline void pack_key_chars(const std::string& k, uint64_t& hi, uint64_t& lo) {
    unsigned char b[16] = {0};
    std::memcpy(b, k.data(), k.size() < 16 ? k.size() : 16);
    hi = 0; lo = 0;
    for (int i = 0; i < 8; ++i) { hi = (hi << 8) | b[i]; lo = (lo << 8) | b[8 + i]; }
}
}  // namespace detail

// Rebuild the reference containers from an export.  `Map` needs map_xy / morton_list / map_cell members of
// the reference's shapes; Node/SlopeT/CellT are its element types.  Without an arena the objects are allocated with `new`
// and owned by the caller, as in the reference (src/receiver.cpp:62,85; map2D.h:598,632,648).
// Rows arrive grouped by column, columns in first-seen order (= morton_list order).  The ordered containers are filled in KEY
// order — the columns' keys are sorted once (as packed integers) and every element goes in with the end() hint, an O(1) insert
// with no string compares — which yields exactly the containers node-by-node insertion builds: std::map order is the key
// order, equal keys of map_xy keep their insertion (= row) order, morton_list is filled in row order.
template <class Map, class Node, class SlopeT, class CellT>
void materialise_into(const gndt_cells& c, Map& out, MapArena<Node, SlopeT, CellT>* arena = nullptr) {
    const uint64_t n = c.num_nodes;
    std::vector<detail::ColumnRun> cols;
    std::vector<std::string> keys;
    cols.reserve(c.num_columns ? c.num_columns : n / 2 + 1);
    keys.reserve(cols.capacity());
    uint64_t n_slopes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (i == 0 || c.sx[i] != c.sx[i - 1] || c.sy[i] != c.sy[i - 1]) {
            keys.push_back(column_key(c.sx[i], c.sy[i]));
            detail::ColumnRun r{0, 0, (uint32_t)i, 0};
            detail::pack_key_chars(keys.back(), r.k_hi, r.k_lo);
            cols.push_back(r);
            out.morton_list.push_back(keys.back());                   // receiver.cpp:70
        }
        ++cols.back().count;
        if (c.flags[i] & GNDT_FLAG_SLOPE) ++n_slopes;
    }
    if (arena) {
        arena->clear();
        arena->nodes.reserve(n); arena->slopes.reserve(n_slopes); arena->cells.reserve(cols.size());
        arena->row_slope.assign(n, nullptr);
    }
    std::vector<uint32_t> order(cols.size());
    for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return cols[a].k_hi != cols[b].k_hi ? cols[a].k_hi < cols[b].k_hi : cols[a].k_lo < cols[b].k_lo;
    });
    for (const uint32_t k : order) {
        const std::string& key = keys[k];
        CellT* cell;
        if (arena) { arena->cells.emplace_back(key); cell = &arena->cells.back(); }
        else cell = new CellT(key);                                   // map2D.h:598-599
        out.map_cell.emplace_hint(out.map_cell.end(), key, cell);
        for (uint64_t i = cols[k].first, e = (uint64_t)cols[k].first + cols[k].count; i < e; ++i) {
            Node* node;
            if (arena) { arena->nodes.emplace_back(); node = &arena->nodes.back(); }
            else node = new Node();
            node->morton = key;
            node->z = c.sz[i];
            node->count = c.count[i];
            if (c.flags[i] & GNDT_FLAG_HAS_STATS) {                    // map2D.h:621-625
                node->N = (int)c.count[i];
                for (int q = 0; q < 3; ++q) node->xyz_centroid(q) = c.mean[3 * i + q];
                const float* u = c.cov + 6 * i;                       // xx,xy,xz,yy,yz,zz
                node->covariance_matrix(0, 0) = u[0]; node->covariance_matrix(0, 1) = u[1]; node->covariance_matrix(0, 2) = u[2];
                node->covariance_matrix(1, 0) = u[1]; node->covariance_matrix(1, 1) = u[3]; node->covariance_matrix(1, 2) = u[4];
                node->covariance_matrix(2, 0) = u[2]; node->covariance_matrix(2, 1) = u[4]; node->covariance_matrix(2, 2) = u[5];
            }
            node->_isSlope = (c.flags[i] & GNDT_FLAG_SLOPE) != 0;
            out.map_xy.emplace_hint(out.map_xy.end(), key, node);     // equal keys keep insertion order
            if (c.flags[i] & GNDT_FLAG_SLOPE) {                       // map2D.h:632-642 / 648-658
                SlopeT* sl;
                if (arena) { arena->slopes.emplace_back(); sl = &arena->slopes.back(); arena->row_slope[i] = sl; }
                else sl = new SlopeT();
                sl->morton_xy = key;
                sl->morton_z = c.sz[i];
                sl->down = (c.flags[i] & GNDT_FLAG_DOWN) != 0;
                sl->h = sl->g = sl->f = FLT_MAX;
                for (int q = 0; q < 3; ++q) { sl->mean(q) = c.mean[3 * i + q]; sl->normal(q) = c.normal[3 * i + q]; }
                sl->father = nullptr;
                sl->rough = c.rough[i];
                cell->map_slope.insert(std::make_pair((int)c.sz[i], sl));
            }
        }
    }
}

// Write the flood's h (gndt_cost_export, one value per result row) into the Slope objects of containers that
// were materialised from the same export: what computeCost leaves behind in Slope::h (map2D.h:1301, 1329, 1340).
// With the arena's row_slope it is one pass over the rows; without, the slopes are looked up by key.
template <class Map>
void apply_cost_into(const gndt_cells& c, const float* h, Map& out) {
    int32_t cur_sx = 0, cur_sy = 0;
    typename decltype(out.map_cell)::iterator cell = out.map_cell.end();
    for (uint64_t i = 0; i < c.num_nodes; ++i) {
        if (i == 0 || c.sx[i] != cur_sx || c.sy[i] != cur_sy) {
            cur_sx = c.sx[i]; cur_sy = c.sy[i];
            cell = out.map_cell.find(column_key(cur_sx, cur_sy));
        }
        if (!(c.flags[i] & GNDT_FLAG_SLOPE) || cell == out.map_cell.end()) continue;
        auto it = cell->second->map_slope.find((int)c.sz[i]);
        if (it != cell->second->map_slope.end()) it->second->h = h[i];
    }
}
template <class SlopeT>
void apply_cost_rows(const std::vector<SlopeT*>& row_slope, const float* h) {
    for (size_t i = 0; i < row_slope.size(); ++i)
        if (row_slope[i]) row_slope[i]->h = h[i];
}

// include/robot.h:12-46 (ROS-free; setPos/setGoal take the parsed vectors)
class RobotSphere {
    float r;
    Vector3f position, goal;
public:
    explicit RobotSphere(float rr) : r(rr) {
        position.d[0] = 30.02865f; position.d[1] = 1.2212f; position.d[2] = 0.40626f;      // robot.h:33-34
        goal.d[0] = 63.02865f; goal.d[1] = -37.2212f; goal.d[2] = 1.3026f;
    }
    RobotSphere(float rr, const Vector3f& pos, const Vector3f& g) : r(rr), position(pos), goal(g) {}
    float getRobotR() const { return r; }
    Vector3f getPosition() const { return position; }
    Vector3f getGoal() const { return goal; }
    float getReachableHeight() const { return 0.15f; }    // robot.h:38-39
    float getRough() const { return 100.f; }              // robot.h:40-43
    float getAngle() const { return 30.f; }               // robot.h:44-46
    void setPos(const Vector3f& p) { position = p; }
    void setGoal(const Vector3f& g) { goal = g; }
};

// daysun::TwoDmap for the build path (include/map2D.h:190-194, 485-507, 592, 950).
class TwoDmap {
    float gridLen, zLen;
    Vector3f cloudFirst;
    float slope_interval = 0.f;
    gndt_handle* handle = nullptr;
    int handle_demand = -1;
    std::string last_error;
    CellsHost host;                     // the export the containers were built from (rows <-> Slope objects)
    gndt_cost_stats cost_stats{};
    MapArena<OcNode, Slope, Cell> arena;   // the containers' objects (eager mode)
    // lazy mode (create2DMap(..., lazy = true)): no containers; the consumers are served from the export by row index
    bool lazy_mode = false;
    struct ColumnSlot { int32_t sx, sy; uint32_t first, count; };     // count == 0: free
    std::vector<ColumnSlot> col_index;                                  // open addressing, power-of-two size
    std::deque<Slope> lazy_pool;                                        // Slope objects made on demand (stable addresses)
    std::vector<Slope*> lazy_row_slope;                                 // row -> its Slope once somebody asked for it
    std::vector<float> cost_h;                                          // the last flood's h per row (empty: none yet)
    static uint32_t col_hash(int32_t sx, int32_t sy) {
        uint32_t h = (uint32_t)sx * 0x9E3779B1u ^ (uint32_t)sy * 0x85EBCA77u;
        return h ^ (h >> 15);
    }
    void build_column_index() {
        const gndt_cells& c = host.view;
        size_t cap = 16;
        while (cap < 2 * (size_t)(c.num_columns ? c.num_columns : c.num_nodes) + 2) cap <<= 1;
        col_index.assign(cap, ColumnSlot{0, 0, 0, 0});
        for (uint64_t i = 0; i < c.num_nodes;) {
            uint64_t e = i + 1;
            while (e < c.num_nodes && c.sx[e] == c.sx[i] && c.sy[e] == c.sy[i]) ++e;
            size_t p = col_hash(c.sx[i], c.sy[i]) & (cap - 1);
            while (col_index[p].count) p = (p + 1) & (cap - 1);
            col_index[p] = ColumnSlot{c.sx[i], c.sy[i], (uint32_t)i, (uint32_t)(e - i)};
            i = e;
        }
    }
    const ColumnSlot* find_column(int32_t sx, int32_t sy) const {
        if (col_index.empty()) return nullptr;
        const size_t cap = col_index.size();
        for (size_t p = col_hash(sx, sy) & (cap - 1); col_index[p].count; p = (p + 1) & (cap - 1))
            if (col_index[p].sx == sx && col_index[p].sy == sy) return &col_index[p];
        return nullptr;
    }
    Slope* lazy_slope(uint32_t row) {
        if (lazy_row_slope[row]) return lazy_row_slope[row];
        const gndt_cells& c = host.view;
        lazy_pool.emplace_back();
        Slope* sl = &lazy_pool.back();
        sl->morton_xy = column_key(c.sx[row], c.sy[row]);
        sl->morton_z = c.sz[row];
        sl->down = (c.flags[row] & GNDT_FLAG_DOWN) != 0;
        for (int q = 0; q < 3; ++q) { sl->mean(q) = c.mean[3 * row + q]; sl->normal(q) = c.normal[3 * row + q]; }
        sl->rough = c.rough[row];
        if (!cost_h.empty()) sl->h = cost_h[row];
        lazy_row_slope[row] = sl;
        return sl;
    }
    static bool key_to_column(const std::string& key, int32_t& sx, int32_t& sy) {
        if (key.size() < 2) return false;
        int32_t x = 0, y = 0;
        gndt_morton_to_xy((int32_t)std::atoi(key.c_str() + 1), &x, &y);
        const char q = key[0];
        sx = (q == 'A' || q == 'B') ? x : -x;
        sy = (q == 'A' || q == 'C') ? y : -y;
        return q >= 'A' && q <= 'D';
    }
    bool finish_build(bool lazy) {
        const auto t0 = std::chrono::steady_clock::now();
        clear();
        lazy_mode = lazy;
        if (lazy) {
            build_column_index();
            lazy_row_slope.assign(host.view.num_nodes, nullptr);
        } else {
            materialise_into<TwoDmap, OcNode, Slope, Cell>(host.view, *this, &arena);
        }
        timing.containers_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return true;
    }
public:
    // where the last create2DMap* spent its time on the host side of the seam (tools/host_path.cpp, bench.py "host_path")
    struct Timing { double build_ms = 0, export_ms = 0, containers_ms = 0; } timing;
    std::multimap<std::string, OcNode*> map_xy;
    std::list<std::string> morton_list;
    std::list<std::string> changeMorton_list, delMorton_list;   // unused by the build path; kept for shape
    std::map<std::string, Cell*> map_cell;
    int device_id = 0;
    uint64_t max_nodes_hint = 0;
    int strategy = GNDT_STRATEGY_AUTO;

    TwoDmap(float res, float zres) : gridLen(res), zLen(zres) {}
    ~TwoDmap() { clear(); if (handle) gndt_destroy(handle); }
    TwoDmap(const TwoDmap&) = delete;
    TwoDmap& operator=(const TwoDmap&) = delete;

    float getGridLen() const { return gridLen; }
    float getZLen() const { return zLen; }
    void setCloudFirst(const Vector3f& p) { cloudFirst = p; }
    void setLen(float len) { gridLen = len; drop_handle(); }
    void setZLen(float len) { zLen = len; drop_handle(); }
    void setInterval(float interval) { slope_interval = interval; drop_handle(); }
    float getInterval() const { return slope_interval; }
    const std::string& lastError() const { return last_error; }

    // include/map2D.h:950-976
    bool transMortonXYZ(const Vector3f& position, std::string& morton_xy, int& morton_z) const {
        char q, key[16];
        int32_t nx, ny, sz;
        const int rc = gndt_trans_morton_xyz(cloudFirst.d, gridLen, zLen, position.d, &q, &nx, &ny, &sz, key);
        morton_xy = key; morton_z = sz;
        return rc == GNDT_OK;
    }

    // No reference counterpart — call it from main() before ros::spin() (src/receiver.cpp:283): the reference builds ONE map per
    // process, so what its user waits for is a process's FIRST build; gndt_warmup loads the kernels' code (a temporary handle builds a
    // synthetic cloud of `expected_points` points through every strategy family) and reserves the buffers a cloud of that size
    // needs, so that the first frame costs what the next one does (0.13 ms instead of 0.4-1.0 on a 200 k-point frame).
    bool warmup(const std::string& demand, uint64_t expected_points) {
        const int d = (demand == "true") ? GNDT_DEMAND_TRUE : GNDT_DEMAND_SLOPE;
        if (!ensure_handle(d)) return false;
        const int rc = gndt_warmup(handle, expected_points);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        return true;
    }

    // Replaces `for (i = 1 .. n-1) uniformDivision(points[i], false); map2D.create2DMap(demand);`
    // (src/receiver.cpp:150-160).  `xyz` = host pointer to point 1 (point 0 was given to setCloudFirst),
    // n = number of points to bin, stride_bytes = 12 or 16 (pcl::PointXYZ).
    // lazy = true: the reference's containers (map_xy, morton_list, map_cell) are NOT filled — rebuilding them costs 100-1000x
    // the GPU build (one heap node per tree entry) — and the consumers of this header (computeCost, AccessibleNeighbors,
    // findSlope, AstarPlanar) are served from the exported rows instead, Slope objects being made only for the rows the
    // planner touches.  Code that walks the containers itself needs lazy = false.
    bool create2DMap(const std::string& demand, const void* xyz, size_t n, size_t stride_bytes, bool lazy = false) {
        const int d = (demand == "true") ? GNDT_DEMAND_TRUE : GNDT_DEMAND_SLOPE;
        if (!ensure_handle(d)) return false;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = gndt_set_origin(handle, cloudFirst.d);
        if (rc == GNDT_OK) rc = gndt_build(handle, xyz, n, stride_bytes);
        uint64_t nodes = 0, cols = 0, slopes = 0;
        if (rc == GNDT_OK) rc = gndt_sync(handle, &nodes, &cols, &slopes);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }   // "wrong" (map2D.h:602-604)
        const auto t1 = std::chrono::steady_clock::now();
        host.resize(0);                                  // (the rows are read where gndt_export_host puts them: the handle's pinned mirror)
        rc = gndt_export_host(handle, &host.view);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        const auto t2 = std::chrono::steady_clock::now();
        timing.build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        timing.export_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        return finish_build(lazy);
    }
    bool isLazy() const { return lazy_mode; }

    // map_cell.find(morton_xy)->second->map_slope.find(morton_z) (map2D.h:1294-1299, GlobalPlan.h:58-64) in either mode
    Slope* findSlope(const std::string& morton_xy, int morton_z) {
        if (!lazy_mode) {
            auto it = map_cell.find(morton_xy);
            if (it == map_cell.end()) return nullptr;
            auto ss = it->second->map_slope.find(morton_z);
            return ss == it->second->map_slope.end() ? nullptr : ss->second;
        }
        int32_t sx, sy;
        if (!key_to_column(morton_xy, sx, sy)) return nullptr;
        const ColumnSlot* col = find_column(sx, sy);
        if (!col) return nullptr;
        const gndt_cells& c = host.view;
        for (uint32_t r = col->first; r < col->first + col->count; ++r)
            if (c.sz[r] == morton_z && (c.flags[r] & GNDT_FLAG_SLOPE)) return lazy_slope(r);
        return nullptr;
    }

    // Replaces TwoDmap::computeCost (include/map2D.h:1285-1397; receiver.cpp:171): the flood runs on the GPU over the
    // grid create2DMap left there, then Slope::h of every slope is set as the reference's loop would leave it.
    // Returns false on an ABI error; goal-lookup failures behave like the reference (nothing is changed).
    bool computeCost(const Vector3f& goal, RobotSphere& robot, const std::string& demand) {
        (void)demand;                                   // the handle already carries the demand of create2DMap
        if (!handle) { last_error = "computeCost before create2DMap"; return false; }
        gndt_robot R{robot.getRobotR(), robot.getReachableHeight(), robot.getRough(), robot.getAngle()};
        int rc = gndt_compute_cost(handle, goal.d, &R, nullptr);
        std::vector<float> h(host.view.num_nodes);
        if (rc == GNDT_OK) rc = gndt_cost_export(handle, h.data(), nullptr, &cost_stats);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        if (cost_stats.goal_status == 0) {
            if (lazy_mode) {                            // slopes made from now on take their h from here; the ones that exist are updated
                cost_h.swap(h);
                for (size_t i = 0; i < lazy_row_slope.size(); ++i) if (lazy_row_slope[i]) lazy_row_slope[i]->h = cost_h[i];
            } else if (arena.row_slope.size() == host.view.num_nodes) apply_cost_rows(arena.row_slope, h.data());
            else apply_cost_into(host.view, h.data(), *this);
        }
        return true;
    }
    const gndt_cost_stats& costStats() const { return cost_stats; }
    const gndt_cells& exported() const { return host.view; }   // the rows the containers were built from

    // include/map2D.h:523-526
    float TravelCost(const Vector3f& cur, const Vector3f& des, float = 0) const {
        const float dx = cur(0) - des(0), dy = cur(1) - des(1), dz = cur(2) - des(2);
        return (float)std::sqrt(((double)dx * (double)dx + (double)dy * (double)dy) + (double)dz * (double)dz);
    }

    // include/map2D.h:477-482
    static float countAngle(const Vector3f& n1, const Vector3f& n2) {
        const float dot = n1(0) * n2(0) + (n1(1) * n2(1) + n1(2) * n2(2));
        const double l1 = std::sqrt(((double)n1(0) * n1(0) + (double)n1(1) * n1(1)) + (double)n1(2) * n1(2));
        const double l2 = std::sqrt(((double)n2(0) * n2(0) + (double)n2(1) * n2(1)) + (double)n2(2) * n2(2));
        const float res = (float)((double)dot / (l1 * l2));
        float an = (float)((double)(std::acos(res) * 180.0f) / 3.14159265358979323846);
        if (an > 90) an = 180 - an;
        return an;
    }

    // Slope::countUp (include/map2D.h:147-177), evaluated against this map's map_xy
    bool countUp(Slope* s) const {
        int zadd = s->morton_z + 1;
        if (s->morton_z == -1) zadd = 1;
        bool zup = false;
        if (lazy_mode) {                                   // the column's nodes in row (= map_xy insertion) order; no statistics: centroid 0
            int32_t sx, sy;
            const ColumnSlot* col = key_to_column(s->morton_xy, sx, sy) ? find_column(sx, sy) : nullptr;
            const gndt_cells& c = host.view;
            for (uint32_t r = col ? col->first : 0; col && r < col->first + col->count && !zup; ++r) {
                const float cz = (c.flags[r] & GNDT_FLAG_HAS_STATS) ? c.mean[3 * r + 2] : 0.f;
                if (zadd == c.sz[r] && std::fabs(cz - s->mean(2)) > slope_interval) zup = true;
            }
            s->up = zup;
            return zup;
        }
        for (auto it = map_xy.find(s->morton_xy); it != map_xy.end() && it->first == s->morton_xy && !zup; ++it)
            if (zadd == it->second->z && std::fabs(it->second->xyz_centroid(2) - s->mean(2)) > slope_interval) zup = true;
        s->up = zup;
        return zup;
    }

    // include/map2D.h:530-548 with countReachable :262-294 and countLRFB :197-259: the accessible slopes of the four
    // neighbouring cells (left, right, forward, back; ascending z inside a cell).  comand: 2.5 planner / collision
    // ring, 3 every slope, 4 3-D planner (evaluates `up` lazily).
    std::list<Slope*> AccessibleNeighbors(Slope* slope, RobotSphere& robot, float comand) {
        std::list<Slope*> list;
        const char q = slope->morton_xy[0];
        int32_t x = 0, y = 0;
        gndt_morton_to_xy((int32_t)std::atoi(slope->morton_xy.c_str() + 1), &x, &y);
        const int sx = (q == 'A' || q == 'B') ? x : -x, sy = (q == 'A' || q == 'C') ? y : -y;
        auto step = [](int v, int d) { int r = v + d; if (r == 0) r += d; return r; };   // no cell 0 (map2D.h:226-255)
        const int nb[4][2] = {{sx, step(sy, -1)}, {sx, step(sy, +1)}, {step(sx, +1), sy}, {step(sx, -1), sy}};
        auto visit = [&](Slope* s) {
            if (comand == 3) { list.push_back(s); return; }
            if (comand == 4) s->up = countUp(s);
            if (s->up) return;
            if (s->rough <= robot.getRough() && countAngle(s->normal, slope->normal) <= robot.getAngle() &&
                std::fabs(s->mean(2) - slope->mean(2)) <= robot.getReachableHeight())
                list.push_back(s);
        };
        for (const auto& c : nb) {
            if (lazy_mode) {                               // the cell's slopes in ascending z, as map_slope iterates them
                const ColumnSlot* col = find_column(c[0], c[1]);
                if (!col) continue;
                uint32_t rows[64];
                std::vector<uint32_t> more;
                uint32_t m = 0;
                for (uint32_t r = col->first; r < col->first + col->count; ++r)
                    if (host.view.flags[r] & GNDT_FLAG_SLOPE) { if (m < 64) rows[m] = r; else more.push_back(r); ++m; }
                if (!more.empty()) { more.insert(more.begin(), rows, rows + 64); }
                uint32_t* rr = more.empty() ? rows : more.data();
                std::sort(rr, rr + m, [&](uint32_t a, uint32_t b) { return host.view.sz[a] < host.view.sz[b]; });
                for (uint32_t k = 0; k < m; ++k) visit(lazy_slope(rr[k]));
                continue;
            }
            auto mit = map_cell.find(column_key(c[0], c[1]));
            if (mit == map_cell.end()) continue;
            for (auto& kv : mit->second->map_slope) visit(kv.second);
        }
        return list;
    }

    // The whole of chatterCallback's front half (src/receiver.cpp:140-160) from a raw cloud: records as a PointCloud2 /
    // .pcd payload lays them out -> NaN rows dropped on the device (publisher.cpp:24-26), origin := the first valid
    // point (receiver.cpp:145), division + create2DMap of the rest.
    bool create2DMapFromRaw(const std::string& demand, const void* raw, size_t n, const gndt_point_layout& layout, bool lazy = false) {
        const int d = (demand == "true") ? GNDT_DEMAND_TRUE : GNDT_DEMAND_SLOPE;
        if (!ensure_handle(d)) return false;
        int rc = gndt_build_cloud(handle, raw, n, &layout);
        uint64_t nodes = 0, cols = 0, slopes = 0;
        if (rc == GNDT_OK) rc = gndt_sync(handle, &nodes, &cols, &slopes);
        if (rc == GNDT_OK) rc = gndt_get_origin(handle, cloudFirst.d);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        host.resize(0);
        rc = gndt_export_host(handle, &host.view);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        return finish_build(lazy);
    }

    // Fill the containers from an export made elsewhere (tests, the sharded build's gathered map): eager mode.
    void adopt(const gndt_cells& c) {
        clear();
        lazy_mode = false;
        if (c.sx == host.view.sx && c.num_nodes == host.view.num_nodes) {      // this map's own export: nothing to copy
            materialise_into<TwoDmap, OcNode, Slope, Cell>(host.view, *this, &arena);
            return;
        }
        const uint64_t k_cols = c.num_columns, k_slopes = c.num_slopes;
        host.resize(c.num_nodes);
        host.view.num_columns = k_cols; host.view.num_slopes = k_slopes;
        const size_t n = c.num_nodes;
        std::memcpy(host.sx.data(), c.sx, 4 * n); std::memcpy(host.sy.data(), c.sy, 4 * n); std::memcpy(host.sz.data(), c.sz, 4 * n);
        std::memcpy(host.count.data(), c.count, 4 * n); std::memcpy(host.first_idx.data(), c.first_idx, 4 * n);
        std::memcpy(host.flags.data(), c.flags, 4 * n); std::memcpy(host.mean.data(), c.mean, 12 * n);
        std::memcpy(host.cov.data(), c.cov, 24 * n); std::memcpy(host.rough.data(), c.rough, 4 * n); std::memcpy(host.normal.data(), c.normal, 12 * n);
        materialise_into<TwoDmap, OcNode, Slope, Cell>(host.view, *this, &arena);
    }

    void clear() {                                         // (the objects live in the arena / the lazy pool, not in `new`s of their own)
        map_xy.clear(); map_cell.clear(); morton_list.clear();
        arena.clear();
        col_index.clear(); lazy_pool.clear(); lazy_row_slope.clear(); cost_h.clear();
    }

private:
    void drop_handle() { if (handle) { gndt_destroy(handle); handle = nullptr; } handle_demand = -1; }
    bool ensure_handle(int demand) {
        if (handle && handle_demand == demand) return true;
        drop_handle();
        gndt_params P{};
        P.grid_len = gridLen; P.z_len = zLen; P.slope_interval = slope_interval;
        P.demand = demand; P.min_points = 3; P.device_id = device_id; P.strategy = strategy;
        P.max_points_hint = 0; P.max_nodes_hint = max_nodes_hint;
        const int rc = gndt_create(&P, &handle);
        if (rc != GNDT_OK) { last_error = gndt_last_error(nullptr); handle = nullptr; return false; }
        handle_demand = demand;
        return true;
    }
};

inline void materialise(const gndt_cells& c, TwoDmap& out) { out.adopt(c); }

// include/GlobalPlan.h:15-166 — the A* planner that consumes Slope::h, restated against these containers with the
// reference's own quirks kept: the open queue is a multimap on f; isContaninedOpen only looks at the entries whose
// key equals the candidate's current f AT THE FRONT of the queue (GlobalPlan.h:32-45), so a slope that is open
// with a larger f is re-inserted (a second queue entry) with g, f and father overwritten unconditionally.
class AstarPlanar {
    std::multimap<float, Slope*> open_queue;
    std::list<Slope*> closed_list;
    Vector3f start, goal;
    static bool same(const Slope* a, const Slope* b) { return a->morton_xy == b->morton_xy && a->morton_z == b->morton_z; }
    bool isContainedClosed(const Slope* s) const {
        for (const Slope* c : closed_list)
            if (same(c, s)) return true;
        return false;
    }
    bool isContaninedOpen(const Slope* s, std::multimap<float, Slope*>::iterator& found) {
        for (auto it = open_queue.begin(); it != open_queue.end(); ++it) {
            if (it->first != s->f) break;
            if (same(it->second, s)) { found = it; return true; }
        }
        return false;
    }
public:
    std::list<Slope*> global_path;
    AstarPlanar(const Vector3f& s, const Vector3f& g) : start(s), goal(g) {}

    bool findRoute(TwoDmap& map2D, RobotSphere& robot, const std::string& demand) {
        std::string morton_xy, g_xy;
        int morton_z = 0, g_z = 0;
        bool route = false;
        map2D.transMortonXYZ(start, morton_xy, morton_z);
        map2D.transMortonXYZ(goal, g_xy, g_z);
        Slope* first = map2D.findSlope(morton_xy, morton_z);         // (map_cell.find / map_slope.find, GlobalPlan.h:58-64)
        if (!first) return false;
        first->g = 0;
        first->f = first->g + first->h;
        open_queue.insert(std::make_pair(first->f, first));
        const float comand = (demand == "true") ? 4.f : 2.5f;
        while (!open_queue.empty()) {
            auto it_open = open_queue.begin();
            Slope* temp = it_open->second;
            if (temp->morton_xy == g_xy && temp->morton_z == g_z) {
                route = true;
                global_path.push_front(temp);
                break;
            }
            std::list<Slope*> nei = map2D.AccessibleNeighbors(temp, robot, comand);
            for (Slope* s : nei) {
                std::multimap<float, Slope*>::iterator found;
                if (isContainedClosed(s) || s->h == FLT_MAX) {
                } else if (isContaninedOpen(s, found)) {
                    const float cand = temp->g + map2D.TravelCost(temp->mean, s->mean);
                    if (s->g > cand) {
                        s->g = cand;
                        s->f = s->g + s->h;
                        s->father = temp;
                        open_queue.erase(found);
                        open_queue.insert(std::make_pair(s->f, s));
                    }
                } else {
                    s->g = temp->g + map2D.TravelCost(temp->mean, s->mean);
                    s->f = s->g + s->h;
                    s->father = temp;
                    open_queue.insert(std::make_pair(s->f, s));
                }
            }
            closed_list.push_back(temp);
            open_queue.erase(it_open);
        }
        if (!route) return false;
        for (Slope* i = global_path.front(); i->father != nullptr; i = global_path.front()) global_path.push_front(i->father);
        return true;
    }
};

}  // namespace gndt_compat
