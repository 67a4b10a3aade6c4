// obca_planner_ref.cpp -- REFERENCE mode of the warm-start search: the reference's Hybrid A* restated step by step (host only, part of libobca_plan.so).
//
// obca_planner.cpp is a re-design (exact polygon clipping against the H-representations the NLP uses, a finer grid for the 6 m bay).  This file follows
// AutonomousParking/hybrid_a_star.jl instead, so that main.jl's own call (main.jl:216-219: point-cloud obstacles, XY 0.3 m / yaw 5 deg / obstacle map 0.1 m)
// can be reproduced: node keying by ROUNDED grid indices (calc_config :456-482, calc_index :413-419), 22 motion commands (calc_motion_inputs :293-302),
// successors by four Euler steps of 0.1 m (calc_next_node :341-393, costs included: switch-back 10, steer change 10 per rad, reverse arcs free), the holonomic
// heuristic with obstacles (a_star.jl:47-128: Dijkstra over the 8-connected grid from the goal, cells within VEHICLE_RADIUS of an obstacle point blocked, the
// priority of a queued cell NOT updated when its cost improves), the analytic Reeds-Shepp expansion tried from EVERY popped node (:165-184, :254-290), the
// collision test of collision_check.jl:40-98 (bubble of radius 2.35 m around the car centre, then the angle-sum point-in-rectangle test per obstacle point)
// and the path assembly of get_final_path :509-536.
// Two things are this library's, not the reference's: the Reeds-Shepp curve is the shortest of ALL 48 words (obca_planner.cpp; the reference's family set is not
// complete, tests/test_planner_cpu.py) and ties in the priority queue are broken by insertion order (DataStructures.PriorityQueue leaves them unspecified).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <limits>
#include "../../include/obca_plan.h"

namespace {

struct RefOpts {
    double xyreso = 0.3, yawreso = 5.0 * M_PI / 180, motion = 0.1, n_steer = 5, max_steer = 0.6, wb = 2.7;      // hybrid_a_star.jl:44-68
    double sb_cost = 10, back_cost = 0, steer_change_cost = 10, steer_cost = 0, h_cost = 1, vehicle_radius = 1.0;
    long max_expansions = 2000000;
};
inline double pi_2_pi(double a) { while (a > M_PI) a -= 2 * M_PI; while (a < -M_PI) a += 2 * M_PI; return a; }      // :325-334
inline long jround(double x) { return (long)std::nearbyint(x); }      // Julia's round(Int64, x): to nearest, ties to even

// ---- collision_check.jl:31-98
const double CB = 1.0, CC = 3.7, CI = 2.0, WBUBBLE_DIST = (CB + CC) / 2 - CB, WBUBBLE_R = (CB + CC) / 2;
const double VRX[5] = {CC, CC, -CB, -CB, CC}, VRY[5] = {-CI / 2, CI / 2, CI / 2, -CI / 2, -CI / 2};
bool rect_free(double ix, double iy, double iyaw, double px, double py) {      // rect_check :57-98 for one obstacle point: false = the point lies inside the car
    const double c = std::cos(-iyaw), s = std::sin(-iyaw), tx = px - ix, ty = py - iy, lx = c * tx - s * ty, ly = s * tx + c * ty;
    double sumangle = 0;
    for (int i = 0; i < 4; i++) {
        const double x1 = VRX[i] - lx, y1 = VRY[i] - ly, x2 = VRX[i + 1] - lx, y2 = VRY[i + 1] - ly, d1 = std::hypot(x1, y1), d2 = std::hypot(x2, y2);
        const double th1 = std::atan2(y1, x1), tty = -std::sin(th1) * x2 + std::cos(th1) * y2;
        double tmp = (x1 * x2 + y1 * y2) / (d1 * d2);
        if (tmp >= 1.0) tmp = 1.0;
        sumangle += tty >= 0 ? std::acos(tmp) : -std::acos(tmp);
    }
    return !(sumangle >= M_PI);
}
struct Cloud {      // the obstacle points; `inrange` of the reference's KD-tree as a uniform bucket grid (same result set, the order does not matter to the test)
    const double *ox, *oy; int n; double x0, y0, cell; int nx, ny; std::vector<std::vector<int>> bucket;
    void build(const double *ox_, const double *oy_, int n_) {
        ox = ox_; oy = oy_; n = n_; cell = WBUBBLE_R;
        double xmin = 1e300, ymin = 1e300, xmax = -1e300, ymax = -1e300;
        for (int i = 0; i < n; i++) { xmin = std::min(xmin, ox[i]); xmax = std::max(xmax, ox[i]); ymin = std::min(ymin, oy[i]); ymax = std::max(ymax, oy[i]); }
        x0 = xmin; y0 = ymin; nx = (int)((xmax - xmin) / cell) + 1; ny = (int)((ymax - ymin) / cell) + 1;
        bucket.assign((size_t)nx * ny, {});
        for (int i = 0; i < n; i++) bucket[(size_t)((int)((oy[i] - y0) / cell)) * nx + (int)((ox[i] - x0) / cell)].push_back(i);
    }
    bool pose_free(double x, double y, double yaw) const {      // one pose of check_collision :40-54
        const double cx = x + WBUBBLE_DIST * std::cos(yaw), cy = y + WBUBBLE_DIST * std::sin(yaw);
        const int bx = (int)std::floor((cx - x0) / cell), by = (int)std::floor((cy - y0) / cell);
        for (int jy = by - 1; jy <= by + 1; jy++) for (int jx = bx - 1; jx <= bx + 1; jx++) {
            if (jx < 0 || jy < 0 || jx >= nx || jy >= ny) continue;
            for (int i : bucket[(size_t)jy * nx + jx])
                if (std::hypot(ox[i] - cx, oy[i] - cy) <= WBUBBLE_R && !rect_free(x, y, yaw, ox[i], oy[i])) return false;
        }
        return true;
    }
};

// ---- a_star.jl:47-128, :209-281: cost-to-go of a disc robot on the 8-connected grid
struct DistPolicy {
    long minx, miny, xw, yw; std::vector<double> pmap;      // pmap[(x - minx) + xw * (y - miny)] (1-based offsets as in the reference), inf where not reached
    double at(long xind, long yind) const { const long a = xind - minx, b = yind - miny; return (a >= 1 && a <= xw && b >= 1 && b <= yw) ? pmap[(size_t)(b - 1) * xw + (a - 1)] : INFINITY; }
    void build(double gx, double gy, const double *ox, const double *oy, int n, double reso, double vr) {
        std::vector<double> sx(n), sy(n);
        double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
        for (int i = 0; i < n; i++) { sx[i] = ox[i] / reso; sy[i] = oy[i] / reso; mnx = std::min(mnx, sx[i]); mxx = std::max(mxx, sx[i]); mny = std::min(mny, sy[i]); mxy = std::max(mxy, sy[i]); }
        minx = jround(mnx); miny = jround(mny); xw = jround(mxx) - minx; yw = jround(mxy) - miny;      // calc_obstacle_map :256-281
        std::vector<char> ob((size_t)xw * yw, 0);
        for (long ix = 1; ix <= xw; ix++) for (long iy = 1; iy <= yw; iy++) {
            const double x = ix + minx, y = iy + miny; double best = 1e300;
            for (int i = 0; i < n; i++) best = std::min(best, std::hypot(sx[i] - x, sy[i] - y));
            if (best <= vr / reso) ob[(size_t)(iy - 1) * xw + (ix - 1)] = 1;
        }
        pmap.assign((size_t)xw * yw, INFINITY);
        struct Cell { long x, y; double cost; };
        std::unordered_map<long, Cell> open, closed;
        typedef std::pair<double, std::pair<long, long>> QE;      // (priority at first insertion, (sequence number, index))
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq; long seq = 0;
        auto index = [&](long x, long y) { return (y - miny) * xw + (x - minx); };      // :251-253
        const long gxi = jround(gx / reso), gyi = jround(gy / reso);
        open[index(gxi, gyi)] = {gxi, gyi, 0.0}; pq.push({0.0, {seq++, index(gxi, gyi)}});
        const double mot[8][3] = {{1, 0, 1}, {0, 1, 1}, {-1, 0, 1}, {0, -1, 1}, {-1, -1, std::sqrt(2.0)}, {-1, 1, std::sqrt(2.0)}, {1, -1, std::sqrt(2.0)}, {1, 1, std::sqrt(2.0)}};
        while (!open.empty() && !pq.empty()) {
            const long cid = pq.top().second.second; pq.pop();
            auto itc = open.find(cid); if (itc == open.end()) continue;
            const Cell cur = itc->second; open.erase(itc); closed[cid] = cur;
            for (int m = 0; m < 8; m++) {
                const Cell nd = {cur.x + (long)mot[m][0], cur.y + (long)mot[m][1], cur.cost + mot[m][2]};
                const long a = nd.x - minx, b = nd.y - miny;
                if (a >= xw || a <= 0 || b >= yw || b <= 0 || ob[(size_t)(b - 1) * xw + (a - 1)]) continue;      // verify_node :209-228
                const long ni = index(nd.x, nd.y);
                if (closed.count(ni)) continue;
                auto io = open.find(ni);
                if (io != open.end()) { if (io->second.cost > nd.cost) io->second.cost = nd.cost; }      // (the queue keeps the priority of the first insertion, :96-101)
                else { open[ni] = nd; pq.push({nd.cost, {seq++, ni}}); }
            }
        }
        for (auto &kv : closed) { const long a = kv.second.x - minx, b = kv.second.y - miny; if (a >= 1 && a <= xw && b >= 1 && b <= yw) pmap[(size_t)(b - 1) * xw + (a - 1)] = kv.second.cost; }      // calc_policy_map :118-128
    }
};

struct RNode { long xind, yind, yawind; bool direction; std::vector<double> x, y, yaw; double steer, cost; long pind; };

}  // namespace

extern "C" {

int obca_plan_reference_hybrid_astar(const double start[3], const double goal[3], int nob, const double *ox, const double *oy, const double *opts,
                                     double *path, int cap, int *expansions) {
    if (!start || !goal || nob < 1 || !ox || !oy || !path || cap < 2) return -1;
    RefOpts o;
    if (opts) { o.xyreso = opts[0]; o.yawreso = opts[1] * M_PI / 180; o.motion = opts[2]; o.n_steer = opts[3]; o.max_steer = opts[4]; o.wb = opts[5]; o.sb_cost = opts[6]; o.back_cost = opts[7];
                o.steer_change_cost = opts[8]; o.steer_cost = opts[9]; o.h_cost = opts[10]; o.vehicle_radius = opts[11]; o.max_expansions = (long)opts[12]; }
    const double sx = start[0], sy = start[1], syaw = pi_2_pi(start[2]), gx = goal[0], gy = goal[1], gyaw = pi_2_pi(goal[2]);
    // calc_config :456-482
    double mnx = 1e300, mny = 1e300, mxx = -1e300, mxy = -1e300;
    for (int i = 0; i < nob; i++) { mnx = std::min(mnx, ox[i]); mxx = std::max(mxx, ox[i]); mny = std::min(mny, oy[i]); mxy = std::max(mxy, oy[i]); }
    const long minx = jround(mnx / o.xyreso), miny = jround(mny / o.xyreso), xw = jround(mxx / o.xyreso) - minx, yw = jround(mxy / o.xyreso) - miny;
    const long minyaw = jround(-M_PI / o.yawreso) - 1;
    auto index = [&](const RNode &n) { return (n.yawind - minyaw) * xw * yw + (n.yind - miny) * xw + (n.xind - minx); };      // calc_index :413-419
    Cloud cloud; cloud.build(ox, oy, nob);
    DistPolicy hdp; hdp.build(gx, gy, ox, oy, nob, o.xyreso, o.vehicle_radius);      // calc_holonomic_with_obstacle_heuristic :422-426
    auto node_free = [&](const RNode &n) { for (size_t i = 0; i < n.x.size(); i++) if (!cloud.pose_free(n.x[i], n.y[i], n.yaw[i])) return false; return true; };
    auto cost_of = [&](const RNode &n) { return n.cost + o.h_cost * hdp.at(n.xind, n.yind); };      // calc_cost :540-552 (distance-policy heuristic only)
    RNode nstart{jround(sx / o.xyreso), jround(sy / o.xyreso), jround(syaw / o.yawreso), true, {sx}, {sy}, {syaw}, 0.0, 0.0, -1};
    RNode ngoal{jround(gx / o.xyreso), jround(gy / o.xyreso), jround(gyaw / o.yawreso), true, {gx}, {gy}, {gyaw}, 0.0, 0.0, -1};
    // calc_motion_inputs :293-302
    std::vector<double> u, dd;
    { std::vector<double> u1{0.0}; const int ns = (int)o.n_steer; for (int i = 1; i <= ns; i++) u1.push_back(o.max_steer / o.n_steer * i); for (int i = 1; i <= ns; i++) u1.push_back(-o.max_steer / o.n_steer * i);
      for (int r = 0; r < 2; r++) for (double v : u1) { u.push_back(v); dd.push_back(r == 0 ? 1.0 : -1.0); } }
    std::unordered_map<long, RNode> open, closed;
    typedef std::pair<double, std::pair<long, long>> QE;
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq; long seq = 0, nexp = 0;
    open[index(nstart)] = nstart; pq.push({cost_of(nstart), {seq++, index(nstart)}});
    const double maxc = std::tan(o.max_steer) / o.wb;
    const int rscap = 20000; std::vector<double> rsp((size_t)3 * rscap); std::vector<int> rsd(rscap);
    bool found = false;
    while (true) {
        if (open.empty() || pq.empty() || nexp >= o.max_expansions) break;
        const long cid = pq.top().second.second; pq.pop();
        auto itc = open.find(cid); if (itc == open.end()) continue;
        RNode current = itc->second; nexp++;
        {   // update_node_with_analystic_expantion :165-184 / analystic_expantion :254-290
            const double s3[3] = {current.x.back(), current.y.back(), current.yaw.back()}, g3[3] = {gx, gy, gyaw};
            char word[8]; double seg[5], tot = 0;
            const int nr = obca_plan_reeds_shepp(s3, g3, 1.0 / maxc, o.motion, rsp.data(), rsd.data(), rscap, word, seg, &tot);
            bool ok = nr >= 2;
            for (int i = 0; ok && i < nr; i++) ok = cloud.pose_free(rsp[3 * i], rsp[3 * i + 1], rsp[3 * i + 2]);
            if (ok) {
                for (int i = 1; i + 1 < nr; i++) { current.x.push_back(rsp[3 * i]); current.y.push_back(rsp[3 * i + 1]); current.yaw.push_back(rsp[3 * i + 2]); }
                const int nw = (int)std::strlen(word); double c = 0;      // calc_rs_path_cost :187-232
                for (int i = 0; i < nw; i++) c += seg[i] >= 0 ? seg[i] : std::fabs(seg[i]) * o.back_cost;
                for (int i = 0; i + 1 < nw; i++) if (seg[i] * seg[i + 1] < 0) c += o.sb_cost;
                for (int i = 0; i < nw; i++) if (word[i] != 'S') c += o.steer_cost * std::fabs(o.max_steer);
                for (int i = 0; i + 1 < nw; i++) { const double a = word[i] == 'R' ? -o.max_steer : (word[i] == 'L' ? o.max_steer : 0.0), b = word[i + 1] == 'R' ? -o.max_steer : (word[i + 1] == 'L' ? o.max_steer : 0.0); c += o.steer_change_cost * std::fabs(b - a); }
                current.cost += c;
                closed[index(ngoal)] = current; found = true; break;
            }
        }
        open.erase(cid); closed[cid] = current;
        for (size_t m = 0; m < u.size(); m++) {      // calc_next_node :341-393
            const double arc_l = o.xyreso; const int nlist = (int)jround(arc_l / o.motion) + 1;
            RNode nd; nd.x.resize(nlist); nd.y.resize(nlist); nd.yaw.resize(nlist);
            nd.x[0] = current.x.back() + dd[m] * o.motion * std::cos(current.yaw.back()); nd.y[0] = current.y.back() + dd[m] * o.motion * std::sin(current.yaw.back());
            nd.yaw[0] = pi_2_pi(current.yaw.back() + dd[m] * o.motion / o.wb * std::tan(u[m]));
            for (int i = 0; i + 1 < nlist; i++) {
                nd.x[i + 1] = nd.x[i] + dd[m] * o.motion * std::cos(nd.yaw[i]); nd.y[i + 1] = nd.y[i] + dd[m] * o.motion * std::sin(nd.yaw[i]);
                nd.yaw[i + 1] = pi_2_pi(nd.yaw[i] + dd[m] * o.motion / o.wb * std::tan(u[m]));
            }
            nd.xind = jround(nd.x.back() / o.xyreso); nd.yind = jround(nd.y.back() / o.xyreso); nd.yawind = jround(nd.yaw.back() / o.yawreso);
            nd.direction = dd[m] > 0;
            double added = nd.direction ? std::fabs(arc_l) : std::fabs(arc_l) * o.back_cost;
            if (nd.direction != current.direction) added += o.sb_cost;
            added += o.steer_cost * std::fabs(u[m]) + o.steer_change_cost * std::fabs(current.steer - u[m]);
            nd.steer = u[m]; nd.cost = current.cost + added; nd.pind = cid;
            // verify_index :305-322
            if (nd.xind - minx >= xw || nd.xind - minx <= 0 || nd.yind - miny >= yw || nd.yind - miny <= 0) continue;
            if (!node_free(nd)) continue;
            const long ni = index(nd);
            if (closed.count(ni) || open.count(ni)) continue;      // (a node already in the open set keeps its first cost and parent, :150-158)
            pq.push({cost_of(nd), {seq++, ni}}); open[ni] = std::move(nd);
        }
    }
    if (expansions) *expansions = (int)nexp;
    if (!found) return 0;
    // get_final_path :509-536
    std::vector<double> rx(1, gx), ry(1, gy), ryaw(1, gyaw);
    long nid = index(ngoal);
    for (;;) {
        auto it = closed.find(nid); if (it == closed.end()) return 0;
        const RNode &n = it->second;
        for (size_t i = n.x.size(); i-- > 0;) { rx.push_back(n.x[i]); ry.push_back(n.y[i]); ryaw.push_back(n.yaw[i]); }
        nid = n.pind;
        if (n.xind == nstart.xind && n.yind == nstart.yind && n.yawind == nstart.yawind) break;
    }
    const int K = (int)rx.size(); if (K > cap) return -1;
    for (int i = 0; i < K; i++) { path[3 * i] = rx[K - 1 - i]; path[3 * i + 1] = ry[K - 1 - i]; path[3 * i + 2] = ryaw[K - 1 - i]; }
    return K;
}

// ---------------------------------------------------------------------------------------------------- QuadcopterNavigation/a_star_3D.jl, restated
int obca_plan_reference_astar3d(const double start[3], const double goal[3], int nob, const double *ox_, const double *oy_, const double *oz_,
                                const double room_min[3], const double room_max[3], double reso, double *path, int cap, int *expansions, double *cost_out) {
    if (!start || !goal || nob < 0 || (nob && (!ox_ || !oy_ || !oz_)) || !room_min || !room_max || !(reso > 0) || !path || cap < 2) return -1;
    const double H_WEIGHT = 1.1, VEHICLE_RADIUS = 2.5;                                  // :31-32
    const long sx = jround(start[0] / reso), sy = jround(start[1] / reso), sz = jround(start[2] / reso);      // :74-75
    const long gx = jround(goal[0] / reso), gy = jround(goal[1] / reso), gz = jround(goal[2] / reso);
    // calc_obstacle_map :193-231 -- obstacle points in grid units (:78-80), the two room corners appended to the lists (:196-198: they become obstacle points too)
    std::vector<double> ox(nob + 2), oy(nob + 2), oz(nob + 2);
    for (int i = 0; i < nob; i++) { ox[i] = ox_[i] / reso; oy[i] = oy_[i] / reso; oz[i] = oz_[i] / reso; }
    ox[nob] = room_min[0]; ox[nob + 1] = room_max[0]; oy[nob] = room_min[1]; oy[nob + 1] = room_max[1]; oz[nob] = room_min[2]; oz[nob + 1] = room_max[2];
    const long minx = jround(*std::min_element(ox.begin(), ox.end())), miny = jround(*std::min_element(oy.begin(), oy.end())), minz = jround(*std::min_element(oz.begin(), oz.end()));
    const long maxx = jround(*std::max_element(ox.begin(), ox.end())), maxy = jround(*std::max_element(oy.begin(), oy.end())), maxz = jround(*std::max_element(oz.begin(), oz.end()));
    const long xw = maxx - minx, yw = maxy - miny, zw = maxz - minz;
    if (xw <= 0 || yw <= 0 || zw <= 0 || (double)xw * yw * zw > 2e8) return -1;
    // obmap[ix][iy][iz] (0-based here) <=> the point (ix + minx, iy + miny, iz + minz); blocked iff the NEAREST obstacle point is within VEHICLE_RADIUS / reso (:221-224).
    // The reference asks a KD-tree for the nearest point of every cell; the same predicate is evaluated here from the points' side: every point blocks the cells of the
    // ball around it (606 k cells x 67 k points would be 4e10 distance evaluations the other way round).
    std::vector<unsigned char> ob((size_t)xw * yw * zw, 0);
    const double rad = VEHICLE_RADIUS / reso; const long ir = (long)std::ceil(rad);
    for (size_t p = 0; p < ox.size(); p++) {
        const long cx = (long)std::floor(ox[p]), cy = (long)std::floor(oy[p]), cz = (long)std::floor(oz[p]);
        for (long x = cx - ir; x <= cx + ir + 1; x++) { if (x < minx || x >= minx + xw) continue;
            for (long y = cy - ir; y <= cy + ir + 1; y++) { if (y < miny || y >= miny + yw) continue;
                for (long z = cz - ir; z <= cz + ir + 1; z++) { if (z < minz || z >= minz + zw) continue;
                    const double dx = x - ox[p], dy = y - oy[p], dz = z - oz[p];
                    if (std::sqrt(dx * dx + dy * dy + dz * dz) <= rad) ob[((size_t)(x - minx) * yw + (size_t)(y - miny)) * zw + (size_t)(z - minz)] = 1;
                } } }
    }
    auto index = [&](long x, long y, long z) -> long { return (y - miny) * xw * zw + (x - minx) * zw + (z - minz); };      // calc_index :189-191
    auto hh = [&](long x, long y, long z) { return std::sqrt((double)(x * x + y * y + z * z)); };                              // h :283-288
    struct N3 { long x, y, z; double cost; long pind; };
    std::unordered_map<long, N3> open, closed;
    typedef std::pair<double, std::pair<long, long>> QE;                                 // (priority, (insertion number, node index))
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
    std::unordered_map<long, double> prio;                                                // current priority of a queued index (pqOpen[ind] = ... replaces it, :136)
    long seq = 0, nexp = 0;
    const long sid = index(sx, sy, sz);
    open[sid] = N3{sx, sy, sz, 0.0, -1};
    { const double pr = 0.0 + H_WEIGHT * hh(sx - gx, sy - gy, sz - gz); pq.push({pr, {seq++, sid}}); prio[sid] = pr; }
    // get_motion_model :157-187: the 26 neighbours in the reference's row order, cost = Euclidean length
    long mot[26][3]; double mc[26]; int nm = 0;
    for (long dx = -1; dx <= 1; dx++) for (long dy = -1; dy <= 1; dy++) for (long dz = -1; dz <= 1; dz++) {
        if (!dx && !dy && !dz) continue;
        mot[nm][0] = dx; mot[nm][1] = dy; mot[nm][2] = dz; mc[nm] = std::sqrt((double)(dx * dx + dy * dy + dz * dz)); nm++;
    }
    bool found = false; long goal_id = -1;
    while (true) {
        if (open.empty()) break;                                                          // "Error: No open set" :100-103
        if (pq.empty()) break;
        const QE top = pq.top(); pq.pop();
        const long cid = top.second.second;
        auto itp = prio.find(cid);
        if (itp == prio.end() || itp->second != top.first || !open.count(cid)) continue;  // a stale entry of an index whose priority was replaced (or that was dequeued already)
        prio.erase(itp);
        const N3 cur = open[cid]; nexp++;
        if (cur.x == gx && cur.y == gy && cur.z == gz) { closed[cid] = cur; found = true; goal_id = cid; break; }      // :109-113
        open.erase(cid); closed[cid] = cur;
        for (int i = 0; i < nm; i++) {
            N3 nd{cur.x + mot[i][0], cur.y + mot[i][1], cur.z + mot[i][2], cur.cost + mc[i], cid};
            if (nd.x - minx >= xw || nd.x - minx <= 0 || nd.y - miny >= yw || nd.y - miny <= 0 || nd.z - minz >= zw || nd.z - minz <= 0) continue;      // :118-123
            // obmap[node.x-minx+1, ...] in the reference's 1-based array = the cell of the point (node.x, node.y, node.z) itself
            if (ob[((size_t)(nd.x - minx) * yw + (size_t)(nd.y - miny)) * zw + (size_t)(nd.z - minz)]) continue;                                    // :126
            const long ni = index(nd.x, nd.y, nd.z);
            if (closed.count(ni)) continue;
            auto ito = open.find(ni);
            const double pr = nd.cost + H_WEIGHT * hh(nd.x - gx, nd.y - gy, nd.z - gz);
            if (ito != open.end()) {
                if (ito->second.cost > nd.cost) { ito->second.cost = nd.cost; ito->second.pind = cid; prio[ni] = pr; pq.push({pr, {seq++, ni}}); }      // :132-139
            } else { open[ni] = nd; prio[ni] = pr; pq.push({pr, {seq++, ni}}); }
        }
    }
    if (expansions) *expansions = (int)nexp;
    if (!found) return 0;
    // get_final_path :233-263
    std::vector<long> rx(1, gx), ry(1, gy), rz(1, gz);
    long nid = goal_id;
    for (;;) {
        auto it = closed.find(nid); if (it == closed.end()) return 0;
        const N3 &n = it->second;
        rx.push_back(n.x); ry.push_back(n.y); rz.push_back(n.z);
        nid = n.pind;
        if (rx.back() == sx && ry.back() == sy && rz.back() == sz) break;
        if (nid < 0) return 0;
    }
    const int K = (int)rx.size(); if (K > cap) return -1;
    for (int i = 0; i < K; i++) { path[3 * i] = rx[K - 1 - i] * reso; path[3 * i + 1] = ry[K - 1 - i] * reso; path[3 * i + 2] = rz[K - 1 - i] * reso; }
    if (cost_out) *cost_out = closed[goal_id].cost;
    return K;
}

}  // extern "C"
