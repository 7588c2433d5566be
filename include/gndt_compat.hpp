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
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
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

// Rebuild the reference containers from an export.  `Map` needs map_xy / morton_list / map_cell members of
// the reference's shapes; Node/SlopeT/CellT are its element types.  Objects are allocated with `new`
// and owned by the caller, as in the reference (src/receiver.cpp:62,85; map2D.h:598,632,648).
template <class Map, class Node, class SlopeT, class CellT>
void materialise_into(const gndt_cells& c, Map& out) {
    std::string cur_key;
    CellT* cell = nullptr;
    int32_t cur_sx = 0, cur_sy = 0;
    for (uint64_t i = 0; i < c.num_nodes; ++i) {
        if (i == 0 || c.sx[i] != cur_sx || c.sy[i] != cur_sy) {      // rows are grouped by column, in morton_list order
            cur_sx = c.sx[i]; cur_sy = c.sy[i];
            cur_key = column_key(cur_sx, cur_sy);
            out.morton_list.push_back(cur_key);                       // receiver.cpp:70
            cell = new CellT(cur_key);                                // map2D.h:598-599
            out.map_cell.insert(typename decltype(out.map_cell)::value_type(cur_key, cell));
        }
        Node* node = new Node();
        node->morton = cur_key;
        node->z = c.sz[i];
        node->count = c.count[i];
        const bool has = (c.flags[i] & GNDT_FLAG_HAS_STATS) != 0;
        if (has) {                                                    // map2D.h:621-625
            node->N = (int)c.count[i];
            for (int k = 0; k < 3; ++k) node->xyz_centroid(k) = c.mean[3 * i + k];
            const float* u = c.cov + 6 * i;                           // xx,xy,xz,yy,yz,zz
            node->covariance_matrix(0, 0) = u[0]; node->covariance_matrix(0, 1) = u[1]; node->covariance_matrix(0, 2) = u[2];
            node->covariance_matrix(1, 0) = u[1]; node->covariance_matrix(1, 1) = u[3]; node->covariance_matrix(1, 2) = u[4];
            node->covariance_matrix(2, 0) = u[2]; node->covariance_matrix(2, 1) = u[4]; node->covariance_matrix(2, 2) = u[5];
        }
        node->_isSlope = (c.flags[i] & GNDT_FLAG_SLOPE) != 0;
        out.map_xy.insert(typename decltype(out.map_xy)::value_type(cur_key, node));   // equal keys keep insertion order
        if (c.flags[i] & GNDT_FLAG_SLOPE) {                           // map2D.h:632-642 / 648-658
            SlopeT* s = new SlopeT();
            s->morton_xy = cur_key;
            s->morton_z = c.sz[i];
            s->down = (c.flags[i] & GNDT_FLAG_DOWN) != 0;
            s->h = s->g = s->f = FLT_MAX;
            for (int k = 0; k < 3; ++k) { s->mean(k) = c.mean[3 * i + k]; s->normal(k) = c.normal[3 * i + k]; }
            s->father = nullptr;
            s->rough = c.rough[i];
            cell->map_slope.insert(std::make_pair((int)c.sz[i], s));
        }
    }
}

// Write the flood's h (gndt_cost_export, one value per result row) into the Slope objects of containers that
// were materialised from the same export: what computeCost leaves behind in Slope::h (map2D.h:1301, 1329, 1340).
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
public:
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

    // Replaces `for (i = 1 .. n-1) uniformDivision(points[i], false); map2D.create2DMap(demand);`
    // (src/receiver.cpp:150-160).  `xyz` = host pointer to point 1 (point 0 was given to setCloudFirst),
    // n = number of points to bin, stride_bytes = 12 or 16 (pcl::PointXYZ).
    bool create2DMap(const std::string& demand, const void* xyz, size_t n, size_t stride_bytes) {
        const int d = (demand == "true") ? GNDT_DEMAND_TRUE : GNDT_DEMAND_SLOPE;
        if (!ensure_handle(d)) return false;
        int rc = gndt_set_origin(handle, cloudFirst.d);
        if (rc == GNDT_OK) rc = gndt_build(handle, xyz, n, stride_bytes);
        uint64_t nodes = 0, cols = 0, slopes = 0;
        if (rc == GNDT_OK) rc = gndt_sync(handle, &nodes, &cols, &slopes);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }   // "wrong" (map2D.h:602-604)
        host.resize(nodes);
        rc = gndt_export(handle, &host.view);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        clear();
        materialise_into<TwoDmap, OcNode, Slope, Cell>(host.view, *this);
        return true;
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
        if (cost_stats.goal_status == 0) apply_cost_into(host.view, h.data(), *this);
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
        for (const auto& c : nb) {
            auto mit = map_cell.find(column_key(c[0], c[1]));
            if (mit == map_cell.end()) continue;
            for (auto& kv : mit->second->map_slope) {
                Slope* s = kv.second;
                if (comand == 3) { list.push_back(s); continue; }
                if (comand == 4) s->up = countUp(s);
                if (s->up) continue;
                if (s->rough <= robot.getRough() && countAngle(s->normal, slope->normal) <= robot.getAngle() &&
                    std::fabs(s->mean(2) - slope->mean(2)) <= robot.getReachableHeight())
                    list.push_back(s);
            }
        }
        return list;
    }

    // The whole of chatterCallback's front half (src/receiver.cpp:140-160) from a raw cloud: records as a PointCloud2 /
    // .pcd payload lays them out -> NaN rows dropped on the device (publisher.cpp:24-26), origin := the first valid
    // point (receiver.cpp:145), division + create2DMap of the rest.
    bool create2DMapFromRaw(const std::string& demand, const void* raw, size_t n, const gndt_point_layout& layout) {
        const int d = (demand == "true") ? GNDT_DEMAND_TRUE : GNDT_DEMAND_SLOPE;
        if (!ensure_handle(d)) return false;
        int rc = gndt_build_cloud(handle, raw, n, &layout);
        uint64_t nodes = 0, cols = 0, slopes = 0;
        if (rc == GNDT_OK) rc = gndt_sync(handle, &nodes, &cols, &slopes);
        if (rc == GNDT_OK) rc = gndt_get_origin(handle, cloudFirst.d);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        host.resize(nodes);
        rc = gndt_export(handle, &host.view);
        if (rc != GNDT_OK) { last_error = gndt_last_error(handle); return false; }
        clear();
        materialise_into<TwoDmap, OcNode, Slope, Cell>(host.view, *this);
        return true;
    }

    void clear() {
        for (auto& kv : map_xy) delete kv.second;
        for (auto& kv : map_cell) {
            for (auto& s : kv.second->map_slope) delete s.second;
            delete kv.second;
        }
        map_xy.clear(); map_cell.clear(); morton_list.clear();
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

inline void materialise(const gndt_cells& c, TwoDmap& out) {
    out.clear();
    materialise_into<TwoDmap, OcNode, Slope, Cell>(c, out);
}

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
        auto it = map2D.map_cell.find(morton_xy);
        if (it == map2D.map_cell.end()) return false;
        auto ss = it->second->map_slope.find(morton_z);
        if (ss == it->second->map_slope.end()) return false;
        ss->second->g = 0;
        ss->second->f = ss->second->g + ss->second->h;
        open_queue.insert(std::make_pair(ss->second->f, ss->second));
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
