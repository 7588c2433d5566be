// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under grid_ndt_amd/ or include/ may include, link or call
// this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// A CPU restatement of the reference's cost-map flood over a finished grid (daysun/grid_ndt), the immediate
// consumer of the grid-build path:
//   include/map2D.h:1285-1397  TwoDmap::computeCost       FIFO label-correcting flood from the goal slope
//   include/map2D.h:351-411    CollisionCheck             ring expansion + clearance test          ("slope")
//   include/map2D.h:414-474    CollisionCheck3D           same with lazily evaluated `up`          ("true")
//   include/map2D.h:530-548, 551-568, 571-588  AccessibleNeighbors (three overloads / 3D variant)
//   include/map2D.h:262-294, 296-318, 321-337  countReachable (comand 2.5 / 3 / 4), checkList form, 3D form
//   include/map2D.h:197-259    countLRFB                  4-neighbour column keys across the quadrant seams
//   include/map2D.h:477-482    countAngle                 angle between normals, folded to [0, 90]
//   include/map2D.h:340-348    isContainedQ               list membership by (morton_xy, morton_z)
//   include/map2D.h:147-177    Slope::countUp             lazily evaluated `up` for demand "true"
//   include/map2D.h:523-526    TravelCost                 Euclidean distance between slope means
//   include/robot.h:38-46      RobotSphere thresholds     reachable height 0.15, rough 100, angle 30
//   include/GlobalPlan.h:15-166 AstarPlanar::findRoute    the A* planner that consumes Slope::h
//
// It works on an EXPORTED grid (the SoA rows of gndt_cells / oracle_export, reference node order) and rebuilds
// the containers the reference walks: map_cell (std::map<string, Cell>), Cell::map_slope (std::map<int, Slope*>,
// ascending z), map_xy (std::multimap<string, node>).  Two executions of the same semantics:
//   mode 0  "as shipped": std::list queues and linear isContainedQ scans (O(Q^2); small maps, CPU baseline)
//   mode 1  the three lists Q / closed / traversability replaced by one "was pushed" flag per slope: a slope
//           is in exactly one of the three lists from its first push on, so the membership test is the same.
//
// PARITY STATUS: "parity unpinned".  The reference has no tests or fixtures for this path; Eigen (Vector3f::dot,
// unpinned version) fixes the summation order of the dot product in countAngle, restated here as
// p0 + (p1 + p2) (Eigen's fixed-size redux unroller); acos is libm's float overload.  The thresholds are
// compared exactly as written, including the mis-parenthesised clearance term at map2D.h:388 / :451
// (`(a < b) + 2*r` is always true).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <list>
#include <map>
#include <string>
#include <vector>

#include "ref_codec.hpp"

namespace {

struct RobotO {   // include/robot.h:12-46
    float r, reachable_height, rough, angle;
};

struct NodeO {    // the part of OcNode that countUp reads (map2D.h:38-57)
    int z;
    float cz;     // xyz_centroid(2): zero until the node had >= 3 points (map2D.h:55, 611-621)
};

struct SlopeO {   // map2D.h:136-146
    float normal[3];
    float rough;
    float mean[3];
    float h, g, f;
    SlopeO* father;
    std::string morton_xy;
    int morton_z;
    bool up, down;
    int row;        // row of the exported grid (not in the reference: for the export back)
    char pushed;    // mode 1 only
};

struct CellO {    // map2D.h:181-187
    std::map<int, SlopeO*> map_slope;   // CmpByKeyUD (map2D.h:31-35) is plain `<`
};

struct CostMap {
    float gridLen, zLen, slope_interval;
    float origin[3];
    bool demand_true;
    std::map<std::string, CellO> map_cell;
    std::multimap<std::string, NodeO> map_xy;
    std::vector<SlopeO*> slopes;
    long check_pushes = 0;
    double min_angle_margin = 1e30, min_height_margin = 1e30;   // distance of the closest gate decision to its threshold
    ~CostMap() { for (SlopeO* s : slopes) delete s; }
};

std::string key_of(int sx, int sy) {   // map2D.h:952-962, 971-972
    const char q = (sx > 0) ? ((sy > 0) ? 'A' : 'B') : ((sy > 0) ? 'C' : 'D');
    return std::string(1, q) + count_morton(std::abs(sx), std::abs(sy));
}

// map2D.h:477-482
float count_angle(CostMap& M, const float n1[3], const float n2[3], float limit) {
    const float dot = n1[0] * n2[0] + (n1[1] * n2[1] + n1[2] * n2[2]);
    const double l1 = std::sqrt(((double)n1[0] * (double)n1[0] + (double)n1[1] * (double)n1[1]) + (double)n1[2] * (double)n1[2]);
    const double l2 = std::sqrt(((double)n2[0] * (double)n2[0] + (double)n2[1] * (double)n2[1]) + (double)n2[2] * (double)n2[2]);
    const float res = (float)((double)dot / (l1 * l2));
    float an = (float)((double)(std::acos(res) * 180.0f) / M_PI);
    if (an > 90) an = 180 - an;
    if (an == an) M.min_angle_margin = std::min(M.min_angle_margin, (double)std::fabs(an - limit));
    return an;
}

// map2D.h:523-526
float travel_cost(const float cur[3], const float des[3]) {
    const float dx = cur[0] - des[0], dy = cur[1] - des[1], dz = cur[2] - des[2];
    return (float)std::sqrt(((double)dx * (double)dx + (double)dy * (double)dy) + (double)dz * (double)dz);
}

// map2D.h:147-177
bool count_up(CostMap& M, SlopeO* s) {
    int zadd = s->morton_z + 1;
    if (s->morton_z == -1) zadd = 1;
    auto it = M.map_xy.find(s->morton_xy);
    bool zup = false;
    while (it != M.map_xy.end()) {
        if (it->first.compare(s->morton_xy) != 0) break;
        if (zup) break;
        if (zadd == it->second.z && std::fabs(it->second.cz - s->mean[2]) > M.slope_interval) zup = true;
        ++it;
    }
    s->up = zup;
    return zup;
}

// map2D.h:197-259
void count_lrfb(const std::string& belongXY, int x, int y, std::string& leftMtn, std::string& rightMtn, std::string& forMtn,
                std::string& backMtn) {
    int leftx = x, rightx = x, forwardx = 0, backx = 0, lefty = 0, righty = 0, forwardy = y, backy = y;
    std::string leftBe = belongXY, rightBe = belongXY, forBe = belongXY, backBe = belongXY;
    if (belongXY == "A") { forwardx = x + 1; backx = x - 1; lefty = y - 1; righty = y + 1; }
    if (belongXY == "B") { forwardx = x + 1; backx = x - 1; lefty = y + 1; righty = y - 1; }
    if (belongXY == "C") { forwardx = x - 1; backx = x + 1; lefty = y - 1; righty = y + 1; }
    if (belongXY == "D") { forwardx = x - 1; backx = x + 1; lefty = y + 1; righty = y - 1; }
    if (x == 1) {
        if (belongXY == "A") { backBe = "C"; backx = 1; }
        if (belongXY == "B") { backBe = "D"; backx = 1; }
        if (belongXY == "C") { forBe = "A"; forwardx = 1; }
        if (belongXY == "D") { forBe = "B"; forwardx = 1; }
    }
    if (y == 1) {
        if (belongXY == "A") { leftBe = "B"; lefty = 1; }
        if (belongXY == "C") { leftBe = "D"; lefty = 1; }
        if (belongXY == "B") { rightBe = "A"; righty = 1; }
        if (belongXY == "D") { rightBe = "C"; righty = 1; }
    }
    leftMtn = leftBe + count_morton(leftx, lefty);
    rightMtn = rightBe + count_morton(rightx, righty);
    forMtn = forBe + count_morton(forwardx, forwardy);
    backMtn = backBe + count_morton(backx, backy);
}

bool gates(CostMap& M, const SlopeO* s, const RobotO& robot, const float normal[3], const float mean[3]) {
    if (s->rough <= robot.rough)
        if (count_angle(M, s->normal, normal, robot.angle) <= robot.angle) {
            const float dz = std::fabs(s->mean[2] - mean[2]);
            M.min_height_margin = std::min(M.min_height_margin, (double)std::fabs(dz - robot.reachable_height));
            if (dz <= robot.reachable_height) return true;
        }
    return false;
}

// map2D.h:262-294: comand 4 (3D plan), 2.5 (collision ring, planner), 3 (3D collision ring: every slope)
void count_reachable(CostMap& M, const std::string& mtn, std::list<SlopeO*>& listm, const RobotO& robot, const float normal[3],
                     const float mean[3], float comand) {
    auto mit = M.map_cell.find(mtn);
    if (mit == M.map_cell.end()) return;
    for (auto& kv : mit->second.map_slope) {
        SlopeO* s = kv.second;
        if (comand == 4) {
            s->up = count_up(M, s);
            if (s->up != true && gates(M, s, robot, normal, mean)) listm.push_back(s);
        } else if (comand == 2.5f) {
            if (s->up != true && gates(M, s, robot, normal, mean)) listm.push_back(s);
        } else if (comand == 3) {
            listm.push_back(s);
        }
    }
}

// map2D.h:296-318 (checkList form) and 321-337 (3D form: no up/down information)
void count_reachable_check(CostMap& M, const std::string& mtn, std::list<SlopeO*>& listm, const RobotO& robot,
                           const float normal[3], const float mean[3], bool form3d) {
    auto mit = M.map_cell.find(mtn);
    if (mit == M.map_cell.end()) return;
    for (auto& kv : mit->second.map_slope) {
        SlopeO* s = kv.second;
        if (form3d || s->up != true) {
            ++M.check_pushes;
            if (gates(M, s, robot, normal, mean)) listm.push_back(s);
        }
    }
}

void neighbour_keys(const SlopeO* slope, std::string& l, std::string& r, std::string& f, std::string& b) {
    const std::string belongXY = slope->morton_xy.substr(0, 1);
    const int morton = std::atoi(slope->morton_xy.substr(1, slope->morton_xy.length() - 1).c_str());   // strToInt
    int x, y;
    morton_to_xy(morton, &x, &y);
    count_lrfb(belongXY, x, y, l, r, f, b);
}

// map2D.h:530-548
std::list<SlopeO*> accessible_neighbors(CostMap& M, SlopeO* slope, const RobotO& robot, float comand) {
    std::list<SlopeO*> list;
    std::string l, r, f, b;
    neighbour_keys(slope, l, r, f, b);
    count_reachable(M, l, list, robot, slope->normal, slope->mean, comand);
    count_reachable(M, r, list, robot, slope->normal, slope->mean, comand);
    count_reachable(M, f, list, robot, slope->normal, slope->mean, comand);
    count_reachable(M, b, list, robot, slope->normal, slope->mean, comand);
    return list;
}

// map2D.h:551-568 and 571-588
std::list<SlopeO*> accessible_neighbors_check(CostMap& M, SlopeO* slope, const RobotO& robot, bool form3d) {
    std::list<SlopeO*> list;
    std::string l, r, f, b;
    neighbour_keys(slope, l, r, f, b);
    count_reachable_check(M, l, list, robot, slope->normal, slope->mean, form3d);
    count_reachable_check(M, r, list, robot, slope->normal, slope->mean, form3d);
    count_reachable_check(M, f, list, robot, slope->normal, slope->mean, form3d);
    count_reachable_check(M, b, list, robot, slope->normal, slope->mean, form3d);
    return list;
}

// map2D.h:340-348
bool is_contained(const SlopeO* s, const std::list<SlopeO*>& Q) {
    for (const SlopeO* q : Q)
        if (s->morton_xy.compare(q->morton_xy) == 0 && s->morton_z == q->morton_z) return true;
    return false;
}

// map2D.h:351-411 (form3d = false) and 414-474 (form3d = true).  true = collide.
bool collision_check(CostMap& M, SlopeO* slope, int n, const RobotO& robot, bool form3d) {
    const float r = robot.r;
    if (form3d ? count_up(M, slope) : slope->up) return true;
    std::list<SlopeO*> nowSlope, addSlope, allSlope;
    allSlope.push_back(slope);
    nowSlope.push_back(slope);
    while (n > 0) {
        for (SlopeO* cur : nowSlope) {
            std::list<SlopeO*> nei = accessible_neighbors(M, cur, robot, form3d ? 3.f : 2.5f);
            for (SlopeO* nb : nei)
                if (!is_contained(nb, allSlope)) { addSlope.push_back(nb); allSlope.push_back(nb); }
        }
        --n;
        nowSlope = addSlope;
        addSlope.clear();
    }
    for (SlopeO* s : allSlope) {
        if (s->mean[2] < slope->mean[2] && (form3d ? count_up(M, s) : s->up)) return true;
        // map2D.h:388: `((a < b) + 2*r)` is a non-zero number, i.e. always true
        if ((s->mean[2] > slope->mean[2]) && (((s->mean[2] < slope->mean[2]) + 2 * r) != 0) &&
            (s->mean[2] - slope->mean[2] > robot.reachable_height))
            return true;
    }
    auto it = M.map_cell.find(slope->morton_xy);
    if (it != M.map_cell.end()) {
        auto ss = it->second.map_slope.find(slope->morton_z);
        if (ss == it->second.map_slope.end()) return false;   // "collide wrong": cannot happen for a slope of the map
        ++ss;
        if (ss != it->second.map_slope.end()) {
            if ((ss->second->mean[2] < slope->mean[2] + 2 * r) && (ss->second->mean[2] - slope->mean[2] > robot.reachable_height))
                return true;
            return false;
        }
        return false;
    }
    return false;   // the reference falls off the end here (undefined); unreachable for a slope of the map
}

// map2D.h:1285-1397.  Returns 0 = flood ran, 1 = no cell at the goal (nothing happens), 2 = cell but no slope at
// the goal's level ("Goal position wrong").
int compute_cost(CostMap& M, const float goal[3], const RobotO& robot, int mode, std::vector<char>& state) {
    const bool form3d = M.demand_true;
    std::list<SlopeO*> Q, closed, traversability;
    const Key gk = trans_key(M.origin, M.gridLen, M.zLen, goal[0], goal[1], goal[2]);
    const std::string morton_xy = std::string(1, gk.quadrant) + count_morton(gk.nx, gk.ny);
    auto it = M.map_cell.find(morton_xy);
    if (it == M.map_cell.end()) return 1;
    auto ss = it->second.map_slope.find(gk.sz);
    if (ss == it->second.map_slope.end()) return 2;
    ss->second->h = 0;
    Q.push_back(ss->second);
    ss->second->pushed = 1;
    const int n = (int)((std::ceil(2 * robot.r / M.gridLen) - 1) / 2);
    auto contained = [&](SlopeO* s) {
        if (mode == 1) return s->pushed != 0;
        return is_contained(s, Q) || is_contained(s, closed) || is_contained(s, traversability);
    };
    while (Q.size() != 0) {
        SlopeO* q = Q.front();
        if (!collision_check(M, q, n, robot, form3d)) {
            std::list<SlopeO*> nei = accessible_neighbors_check(M, q, robot, form3d);
            for (SlopeO* nb : nei) {
                if (!form3d && nb->up == true) {           // map2D.h:1316-1318 (dead: such slopes are never listed)
                    nb->h = FLT_MAX;
                    closed.push_back(nb);
                    nb->pushed = 1;
                    state[nb->row] = 2;
                } else {
                    const float cand = q->h + travel_cost(q->mean, nb->mean);
                    if (nb->h > cand) {
                        nb->h = cand;
                        if (!contained(nb)) { Q.push_back(nb); nb->pushed = 1; }
                    }
                }
            }
            traversability.push_back(q);
            state[q->row] = 1;
        } else {
            q->h = FLT_MAX;
            closed.push_back(q);
            state[q->row] = 2;
        }
        Q.pop_front();
    }
    return 0;
}

// include/GlobalPlan.h:49-166.  Returns the path (goal last) as slope pointers; empty = "not find the road".
struct Astar {
    std::multimap<float, SlopeO*> open_queue;     // MyCompare is plain `<` (GlobalPlan.h:9-13)
    std::list<SlopeO*> closed_list;
    std::list<SlopeO*> global_path;

    static bool same(const SlopeO* a, const SlopeO* b) { return a->morton_xy.compare(b->morton_xy) == 0 && a->morton_z == b->morton_z; }
    bool contained_closed(const SlopeO* s) const {            // GlobalPlan.h:20-28
        for (const SlopeO* c : closed_list)
            if (same(c, s)) return true;
        return false;
    }
    // GlobalPlan.h:30-45: walks from the FRONT of the queue and stops at the first key that differs from s->f
    bool contained_open(const SlopeO* s, std::multimap<float, SlopeO*>::iterator& iTemp) {
        auto it = open_queue.begin();
        while (it != open_queue.end()) {
            if (it->first != s->f) break;
            if (same(it->second, s)) { iTemp = it; return true; }
            ++it;
        }
        return false;
    }

    bool find_route(CostMap& M, const float start[3], const float goal[3], const RobotO& robot) {
        const Key sk = trans_key(M.origin, M.gridLen, M.zLen, start[0], start[1], start[2]);
        const Key gk = trans_key(M.origin, M.gridLen, M.zLen, goal[0], goal[1], goal[2]);
        const std::string morton_xy = std::string(1, sk.quadrant) + count_morton(sk.nx, sk.ny);
        const std::string g_xy = std::string(1, gk.quadrant) + count_morton(gk.nx, gk.ny);
        bool route = false;
        auto it = M.map_cell.find(morton_xy);
        if (it == M.map_cell.end()) return false;
        auto ss = it->second.map_slope.find(sk.sz);
        if (ss == it->second.map_slope.end()) return false;
        ss->second->g = 0;
        ss->second->f = ss->second->g + ss->second->h;
        open_queue.insert(std::make_pair(ss->second->f, ss->second));
        while (open_queue.size() != 0) {
            auto it_Open = open_queue.begin();
            SlopeO* temp = it_Open->second;
            if (temp->morton_xy.compare(g_xy) == 0 && temp->morton_z == gk.sz) {
                route = true;
                global_path.push_front(temp);
                break;
            }
            float tmp = 2.5f;
            if (M.demand_true) tmp = 4;
            std::list<SlopeO*> nei = accessible_neighbors(M, temp, robot, tmp);
            for (SlopeO* s : nei) {
                std::multimap<float, SlopeO*>::iterator itTemp;
                if (contained_closed(s) || s->h == FLT_MAX) {
                    // do nothing
                } else if (contained_open(s, itTemp)) {
                    if (s->g > temp->g + travel_cost(temp->mean, s->mean)) {
                        s->g = temp->g + travel_cost(temp->mean, s->mean);
                        s->f = s->g + s->h;
                        s->father = temp;
                        open_queue.erase(itTemp);
                        open_queue.insert(std::make_pair(s->f, s));
                    }
                } else {
                    s->g = temp->g + travel_cost(temp->mean, s->mean);
                    s->f = s->g + s->h;
                    s->father = temp;
                    open_queue.insert(std::make_pair(s->f, s));
                }
            }
            closed_list.push_back(temp);
            open_queue.erase(it_Open);
        }
        if (!route) return false;
        SlopeO* i = global_path.front();
        while (i->father != nullptr) {
            global_path.push_front(i->father);
            i = global_path.front();
        }
        return true;
    }
};

void load_map(CostMap& M, size_t n, const int32_t* sx, const int32_t* sy, const int32_t* sz, const float* mean,
              const float* normal, const float* rough, const uint32_t* flags, const float origin[3], float grid_len,
              float z_len, float slope_interval, int demand_true) {
    M.gridLen = grid_len; M.zLen = z_len; M.slope_interval = slope_interval; M.demand_true = demand_true != 0;
    for (int k = 0; k < 3; ++k) M.origin[k] = origin[k];
    for (size_t i = 0; i < n; ++i) {
        const std::string key = key_of(sx[i], sy[i]);
        NodeO nd;
        nd.z = sz[i];
        nd.cz = (flags[i] & 1u) ? mean[3 * i + 2] : 0.f;
        M.map_xy.insert(std::make_pair(key, nd));
        M.map_cell[key];       // create2DMap makes a Cell for every key of morton_list, slopes or not (map2D.h:598-600)
        if (flags[i] & 2u) {   // a Slope object exists (map2D.h:632, 648)
            SlopeO* s = new SlopeO();
            for (int k = 0; k < 3; ++k) { s->normal[k] = normal[3 * i + k]; s->mean[k] = mean[3 * i + k]; }
            s->rough = rough[i];
            s->h = s->g = s->f = FLT_MAX;     // map2D.h:637
            s->father = nullptr;
            s->morton_xy = key;
            s->morton_z = sz[i];
            s->up = false;                       // value-initialised, never assigned in "slope" (map2D.h:636)
            s->down = (flags[i] & 4u) != 0;
            s->row = (int)i;
            s->pushed = 0;
            M.slopes.push_back(s);
            M.map_cell[key].map_slope[sz[i]] = s;
        }
    }
}

}  // namespace

extern "C" {

// Rows are the exported grid in reference order.  Outputs: h[n] (FLT_MAX where no slope / never reached),
// state[n] (0 untouched, 1 traversable, 2 closed), stats[4] = {traversable, closed, checkList pushes, ring depth n},
// margins[2] = {min |angle - limit| in degrees, min ||dz| - reachable| in metres} over every gate evaluated.
// With `start` non-null the planner runs afterwards (receiver.cpp:171-175): path_rows[0..*path_len) = the rows of
// global_path from the start slope to the goal slope (*path_len = 0: no route).
int oracle_compute_cost(size_t n, const int32_t* sx, const int32_t* sy, const int32_t* sz, const uint32_t* count,
                        const float* mean, const float* normal, const float* rough, const uint32_t* flags,
                        const float origin[3], float grid_len, float z_len, float slope_interval, int demand_true,
                        const float goal[3], const float robot4[4], int mode, float* h_out, uint8_t* state_out,
                        int64_t stats[4], double margins[2], const float* start, int32_t* path_rows, int64_t path_cap,
                        int64_t* path_len) {
    (void)count;
    CostMap M;
    load_map(M, n, sx, sy, sz, mean, normal, rough, flags, origin, grid_len, z_len, slope_interval, demand_true);
    RobotO robot{robot4[0], robot4[1], robot4[2], robot4[3]};
    std::vector<char> state(n, 0);
    const int rc = compute_cost(M, goal, robot, mode, state);
    for (size_t i = 0; i < n; ++i) { h_out[i] = FLT_MAX; state_out[i] = (uint8_t)state[i]; }
    int64_t trav = 0, closed = 0;
    for (SlopeO* s : M.slopes) h_out[s->row] = s->h;
    for (size_t i = 0; i < n; ++i) { trav += state[i] == 1; closed += state[i] == 2; }
    stats[0] = trav; stats[1] = closed; stats[2] = M.check_pushes;
    stats[3] = (int64_t)((std::ceil(2 * robot.r / grid_len) - 1) / 2);
    margins[0] = M.min_angle_margin; margins[1] = M.min_height_margin;
    if (start && path_len) {
        *path_len = 0;
        Astar planner;
        if (planner.find_route(M, start, goal, robot)) {
            int64_t k = 0;
            for (SlopeO* s : planner.global_path) { if (k < path_cap) path_rows[k] = s->row; ++k; }
            *path_len = k;
        }
    }
    return rc;
}

}  // extern "C"
