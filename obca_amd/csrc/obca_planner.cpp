// obca_planner.cpp -- host-side warm-start planner (C ABI, no GPU): a compact Hybrid A* over (x, y, yaw) for the car of the parking
// scenarios.  It plays the role of the step BEFORE the hot path in the reference (SURVEY.md section 8f "next-2"):
//   /root/reference/AutonomousParking/hybrid_a_star.jl (search over motion primitives, grid-based holonomic heuristic,
//   collision check of the car rectangle against the obstacles), main.jl:216-248 (path -> rx, ry, ryaw -> down-sampling).
// This is a re-design, not a port: obstacles are the same convex H-representations the NLP uses (A p <= b per obstacle, obstHrep.jl),
// the collision test is the overlap area of the inflated car rectangle with each obstacle (the rectangle clipped by the obstacle's
// half-planes: exact for convex sets, no KD-tree of sampled obstacle points; separating-axis tests settle nearly every call before
// the clip), the heuristic is max(obstacle-aware 2-D Dijkstra distance of a disc robot, turning-radius bound on the heading error)
// and the analytic Reeds-Shepp expansion (hybrid_a_star.jl:193-214, reeds_shepp.jl) ends the search on the exact goal pose.
// (obca_planner_ref.cpp is the REFERENCE mode: hybrid_a_star.jl restated on its point-cloud obstacles.)
// Cost per expanded node (config-3 bay, 0.1 m / 3 deg cells): ~3.3 us on one core -- 1.4 us the 44 Reeds-Shepp family evaluations
// (which share one sin / cos pair and two polar forms per mirror image), the rest ~26 collision tests, libm and the heap.
// Build: g++ -O2 -shared -fPIC -pthread -o libobca_plan.so obca_planner.cpp obca_planner_ref.cpp   (tools/build.sh, __graft_entry__.build()).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <queue>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <atomic>
#include <thread>
#include "../../include/obca_plan.h"

namespace {

struct P2 { double x, y; };

struct World {
    int nOb; std::vector<int> v, off; std::vector<double> A, b;   // obstacle j: rows off[j] .. off[j]+v[j]
    double xmin, xmax, ymin, ymax;                               // XYbounds of the rear-axle position
    double ego[4], margin;                                        // front, left, rear, right extents from the rear axle; inflation
    mutable std::vector<P2> bufA, bufB;                           // clip scratch of collides(), sized for the obstacle with the most rows
    std::vector<double> rn;                                       // 1 / |a_i| of every row
    std::vector<P2> vert; std::vector<int> voff;                  // obstacle j as a polygon (cut to a box far outside the bounds if it is unbounded): vert[voff[j] .. voff[j+1])
    void finish();
};

// convex polygon clipped by the half-plane a.p <= bb (Sutherland-Hodgman); returns the number of vertices left
static int clip(const P2 *in, int n, double ax, double ay, double bb, P2 *out) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        const P2 &p = in[i], &q = in[(i + 1) % n];
        const double dp = ax * p.x + ay * p.y - bb, dq = ax * q.x + ay * q.y - bb;
        if (dp <= 0) out[m++] = p;
        if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) { const double t = dp / (dp - dq); out[m++] = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)}; }
    }
    return m;
}

void World::finish() {     // what collides() needs besides the rows: their norms, the obstacles' vertices, the clip scratch
    int vmax = 0; for (int j = 0; j < nOb; j++) vmax = std::max(vmax, v[j]);
    bufA.resize(4 + vmax + 1); bufB.resize(4 + vmax + 1);         // every half-plane adds at most one vertex to a convex polygon
    rn.resize(off[nOb]);
    for (int r = 0; r < off[nOb]; r++) { const double n = std::hypot(A[2 * r], A[2 * r + 1]); rn[r] = n > 0 ? 1.0 / n : 0.0; }
    const double far_ = 1e3 + std::max(std::max(std::fabs(xmin), std::fabs(xmax)), std::max(std::fabs(ymin), std::fabs(ymax)));
    voff.assign(nOb + 1, 0); vert.clear();
    std::vector<P2> pa(4 + vmax + 1), pb(4 + vmax + 1);
    for (int j = 0; j < nOb; j++) {
        int n = 4; pa[0] = {far_, far_}; pa[1] = {-far_, far_}; pa[2] = {-far_, -far_}; pa[3] = {far_, -far_};
        P2 *cur = pa.data(), *nxt = pb.data();
        for (int i = 0; i < v[j] && n > 0; i++) { const int r = off[j] + i; n = clip(cur, n, A[2 * r], A[2 * r + 1], b[r], nxt); std::swap(cur, nxt); }
        vert.insert(vert.end(), cur, cur + n); voff[j + 1] = (int)vert.size();
    }
}

// Does the (inflated) car rectangle at the pose overlap an obstacle?  The answer is DEFINED by the clipped overlap area (car cut by every half-plane of the obstacle,
// area > 1e-9); nearly every call is settled before that by the separating-axis tests of a rectangle against a convex polygon, which agree with it: a row all four corners
// violate (the clip would leave nothing), a car side every vertex of the obstacle lies beyond, or a corner deep inside every row.
static bool collides_cs(const World &w, double x, double y, double c, double s) {      // c, s = cos, sin of the heading
    if (x < w.xmin || x > w.xmax || y < w.ymin || y > w.ymax) return true;
    const double m = w.margin;
    const double f = w.ego[0] + m, l = w.ego[1] + m, r = w.ego[2] + m, rt = w.ego[3] + m;
    const P2 car[4] = {{x + f * c - l * s, y + f * s + l * c}, {x - r * c - l * s, y - r * s + l * c},
                       {x - r * c + rt * s, y - r * s - rt * c}, {x + f * c + rt * s, y + f * s - rt * c}};
    std::vector<P2> &bufA = w.bufA, &bufB = w.bufB;
    for (int j = 0; j < w.nOb; j++) {
        bool apart = false; double deep[4] = {1e300, 1e300, 1e300, 1e300};      // deep[q]: how far corner q is inside the obstacle (least over its rows)
        for (int i = 0; i < w.v[j]; i++) {
            const int rix = w.off[j] + i; const double ax = w.A[2 * rix], ay = w.A[2 * rix + 1], bb = w.b[rix];
            const double d0 = ax * car[0].x + ay * car[0].y - bb, d1 = ax * car[1].x + ay * car[1].y - bb, d2 = ax * car[2].x + ay * car[2].y - bb, d3 = ax * car[3].x + ay * car[3].y - bb;
            if (d0 > 0 && d1 > 0 && d2 > 0 && d3 > 0) { apart = true; break; }
            const double n_ = w.rn[rix];
            deep[0] = std::min(deep[0], -d0 * n_); deep[1] = std::min(deep[1], -d1 * n_); deep[2] = std::min(deep[2], -d2 * n_); deep[3] = std::min(deep[3], -d3 * n_);
        }
        if (apart) continue;
        if (std::max(std::max(deep[0], deep[1]), std::max(deep[2], deep[3])) > 1e-3) return true;   // a quarter disc of radius 1e-3 around that corner is overlap: area > 7e-7
        {   // the car's own sides: in its frame the obstacle's vertices are (u, v_); a side with every vertex beyond it (by more than a rounding margin) separates
            const int v0 = w.voff[j], v1 = w.voff[j + 1];
            double umin = 1e300, umax = -1e300, wmin = 1e300, wmax = -1e300;
            for (int q = v0; q < v1; q++) {
                const double dx = w.vert[q].x - x, dy = w.vert[q].y - y, u = c * dx + s * dy, v_ = -s * dx + c * dy;
                umin = std::min(umin, u); umax = std::max(umax, u); wmin = std::min(wmin, v_); wmax = std::max(wmax, v_);
            }
            if (v1 > v0 && (umin > f + 1e-7 || umax < -r - 1e-7 || wmin > l + 1e-7 || wmax < -rt - 1e-7)) continue;
        }
        int n = 4; std::memcpy(bufA.data(), car, sizeof car);
        P2 *cur = bufA.data(), *nxt = bufB.data();
        for (int i = 0; i < w.v[j] && n > 0; i++) {
            const int rix = w.off[j] + i;
            n = clip(cur, n, w.A[2 * rix], w.A[2 * rix + 1], w.b[rix], nxt);
            std::swap(cur, nxt);
        }
        if (n >= 3) {   // non-degenerate overlap
            double area = 0; for (int i = 0; i < n; i++) { const P2 &p = cur[i], &q = cur[(i + 1) % n]; area += p.x * q.y - q.x * p.y; }
            if (std::fabs(area) > 1e-9) return true;
        }
    }
    return false;
}
static inline bool collides(const World &w, double x, double y, double yaw) { return collides_cs(w, x, y, std::cos(yaw), std::sin(yaw)); }

static bool disc_collides(const World &w, double x, double y, double rad) {
    for (int j = 0; j < w.nOb; j++) {   // distance of the point to the convex set (intersection of half-planes), conservative via max row slack
        double worst = -1e300;
        for (int i = 0; i < w.v[j]; i++) { const int r = w.off[j] + i; const double n = std::hypot(w.A[2 * r], w.A[2 * r + 1]); worst = std::max(worst, (w.A[2 * r] * x + w.A[2 * r + 1] * y - w.b[r]) / n); }
        if (worst < rad) {   // inside the rad-offset of every half-plane: within rad of the set unless near a corner (then slightly conservative)
            if (worst <= 0) return true;
            // corner case: exact distance to the set is >= worst; test the rounded corner with the two most violated rows
            int cnt = 0; double d2 = 0;
            for (int i = 0; i < w.v[j]; i++) { const int r = w.off[j] + i; const double n = std::hypot(w.A[2 * r], w.A[2 * r + 1]); const double e = (w.A[2 * r] * x + w.A[2 * r + 1] * y - w.b[r]) / n; if (e > 0) { d2 += e * e; cnt++; } }
            if (cnt <= 1 || d2 < rad * rad) return true;
        }
    }
    return false;
}

struct Node { double x, y, yaw, g; int parent; int8_t dir, steer; };
static inline double wrap(double a) { while (a > M_PI) a -= 2 * M_PI; while (a < -M_PI) a += 2 * M_PI; return a; }

}  // namespace

// ---------------------------------------------------------------- Reeds-Shepp curves (the planner's analytic expansion)
// Shortest path of a car that drives forwards and backwards with a bounded turning radius (Reeds & Shepp 1990), in the normalised problem
// (unit radius, start at the origin with heading 0): the 48 candidate words are generated from the closed-form families CSC, CCC, CCCC, CCSC,
// CCSCC and the time-flip / reflection / backwards symmetries, the shortest is kept.  Stands where hybrid_a_star.jl:262-300 calls
// reeds_shepp.calc_shortest_path (reeds_shepp.jl).  A path is at most five segments: type 'L' / 'R' (arc, signed angle) or 'S' (signed length).
namespace rs {
struct Path { char type[5]; double len[5]; int n; double total; };
static const double PI = M_PI, ZERO = 1e-12;
static inline double mod2pi(double x) { double v = std::fabs(x) < 2 * PI ? x : std::fmod(x, 2 * PI); if (v < -PI) v += 2 * PI; else if (v > PI) v -= 2 * PI; return v; }   // (fmod returns such an x itself)
static inline void polar(double x, double y, double &r, double &th) { r = std::hypot(x, y); th = std::atan2(y, x); }
static inline void tau_omega(double u, double v, double xi, double eta, double phi, double &tau, double &omega) {
    const double delta = mod2pi(u - v), A = std::sin(u) - std::sin(delta), B = std::cos(u) - std::cos(delta) - 1.0;
    const double t1 = std::atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2.0 * (std::cos(delta) - std::cos(v) - std::cos(u)) + 3.0;
    tau = t2 < 0 ? mod2pi(t1 + PI) : mod2pi(t1);
    omega = mod2pi(tau - u + v - phi);
}
// One mirror image of the query, (x, y, phi), with what the families share: sin / cos of phi, the two centre offsets (xi, eta) = (x -+ sin phi, y - 1 +- cos phi) and
// their polar forms.  A family is a few lines on top of these; evaluating the 44 family calls of a query from scratch costs ~200 libm calls, from here ~50.
struct Pre {
    double x, y, phi, sp, cp;
    double xm, em, rm, tm;        // (x - sin phi, y - 1 + cos phi) and its polar form         (LSL, LRL, LRSL)
    double xp, ep, rp, tp;        // (x + sin phi, y - 1 - cos phi) and its polar form         (LSR, LRLR, LRSR, LRSLR)
};
static inline Pre make_pre(double x, double y, double phi, double sp, double cp, bool polar_p) {
    Pre q; q.x = x; q.y = y; q.phi = phi; q.sp = sp; q.cp = cp;
    q.xm = x - sp; q.em = y - 1.0 + cp; polar(q.xm, q.em, q.rm, q.tm);
    q.xp = x + sp; q.ep = y - 1.0 - cp; q.rp = std::hypot(q.xp, q.ep); q.tp = polar_p && q.rp >= 2.0 ? std::atan2(q.ep, q.xp) : 0;      // (the angle: LSR only, which needs rp >= 2)
    return q;
}
static bool LpSpLp(const Pre &q, double &t, double &u, double &v) {
    u = q.rm; t = q.tm;
    if (t >= -ZERO) { v = mod2pi(q.phi - t); if (v >= -ZERO) return true; }
    return false;
}
static bool LpSpRp(const Pre &q, double &t, double &u, double &v) {
    double u1 = q.rp; const double t1 = q.tp;
    u1 *= u1;
    if (u1 >= 4.0) { u = std::sqrt(u1 - 4.0); const double th = std::atan2(2.0, u); t = mod2pi(t1 + th); v = mod2pi(t - q.phi); return t >= -ZERO && v >= -ZERO; }
    return false;
}
static bool LpRmL(const Pre &q, double &t, double &u, double &v) {
    const double u1 = q.rm, th = q.tm;
    if (u1 <= 4.0) { u = -2.0 * std::asin(0.25 * u1); t = mod2pi(th + 0.5 * u + PI); v = mod2pi(q.phi - t + u); return t >= -ZERO && u <= ZERO; }
    return false;
}
static bool LpRupLumRm(const Pre &q, double &t, double &u, double &v) {
    const double rho = 0.25 * (2.0 + q.rp);
    if (rho <= 1.0) { u = std::acos(rho); tau_omega(u, -u, q.xp, q.ep, q.phi, t, v); return t >= -ZERO && v <= ZERO; }
    return false;
}
static bool LpRumLumRp(const Pre &q, double &t, double &u, double &v) {
    const double rho = (20.0 - q.xp * q.xp - q.ep * q.ep) / 16.0;
    if (rho >= 0 && rho <= 1) { u = -std::acos(rho); if (u >= -0.5 * PI) { tau_omega(u, u, q.xp, q.ep, q.phi, t, v); return t >= -ZERO && v >= -ZERO; } }
    return false;
}
static bool LpRmSmLm(const Pre &q, double &t, double &u, double &v) {
    const double rho = q.rm, th = q.tm;
    if (rho >= 2.0) { const double r = std::sqrt(rho * rho - 4.0); u = 2.0 - r; t = mod2pi(th + std::atan2(r, -2.0)); v = mod2pi(q.phi - 0.5 * PI - t); return t >= -ZERO && u <= ZERO && v <= ZERO; }
    return false;
}
static bool LpRmSmRm(const Pre &q, double &t, double &u, double &v) {
    const double rho = q.rp;                                       // polar form of (-eta, xi): the same radius, the angle atan2(xi, -eta) -- negative (a reject) for xi < 0
    if (rho >= 2.0 && !(q.xp < -1e-9 * (1.0 + std::fabs(q.ep)))) { t = std::atan2(q.xp, -q.ep); u = 2.0 - rho; v = mod2pi(t + 0.5 * PI - q.phi); return t >= -ZERO && u <= ZERO && v <= ZERO; }
    return false;
}
static bool LpRmSLmRp(const Pre &q, double &t, double &u, double &v) {
    const double xi = q.xp, eta = q.ep, rho = q.rp;
    if (rho >= 2.0) {
        u = 4.0 - std::sqrt(rho * rho - 4.0);
        if (u <= ZERO) { t = mod2pi(std::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta)); v = mod2pi(t - q.phi); return t >= -ZERO && v >= -ZERO; }
    }
    return false;
}
static void offer(Path &best, const char *ty, int n, const double *len) {
    double tot = 0; for (int i = 0; i < n; i++) tot += std::fabs(len[i]);
    if (tot < best.total) { best.n = n; best.total = tot; for (int i = 0; i < n; i++) { best.type[i] = ty[i]; best.len[i] = len[i]; } }
}
// every family is tried on (x, y, phi), its time flip (-x, y, -phi: all lengths negated), its reflection (x, -y, -phi: L <-> R) and both: q[0..3]
template <class F> static void four(Path &best, F f, const Pre *q, const char *ty, const char *tyr, int n, void (*fill)(double, double, double, double *)) {
    double t, u, v, len[5];
    if (f(q[0], t, u, v)) { fill(t, u, v, len); offer(best, ty, n, len); }
    if (f(q[1], t, u, v)) { fill(t, u, v, len); for (int i = 0; i < n; i++) len[i] = -len[i]; offer(best, ty, n, len); }
    if (f(q[2], t, u, v)) { fill(t, u, v, len); offer(best, tyr, n, len); }
    if (f(q[3], t, u, v)) { fill(t, u, v, len); for (int i = 0; i < n; i++) len[i] = -len[i]; offer(best, tyr, n, len); }
}
static void f_tuv(double t, double u, double v, double *l) { l[0] = t; l[1] = u; l[2] = v; }
static void f_vut(double t, double u, double v, double *l) { l[0] = v; l[1] = u; l[2] = t; }
static void f_tuuv_m(double t, double u, double v, double *l) { l[0] = t; l[1] = u; l[2] = -u; l[3] = v; }
static void f_tuuv_p(double t, double u, double v, double *l) { l[0] = t; l[1] = u; l[2] = u; l[3] = v; }
static void f_t_q_u_v(double t, double u, double v, double *l) { l[0] = t; l[1] = -0.5 * PI; l[2] = u; l[3] = v; }
static void f_v_u_q_t(double t, double u, double v, double *l) { l[0] = v; l[1] = u; l[2] = -0.5 * PI; l[3] = t; }
static void f_t_q_u_q_v(double t, double u, double v, double *l) { l[0] = t; l[1] = -0.5 * PI; l[2] = u; l[3] = -0.5 * PI; l[4] = v; }
static Path shortest(double x, double y, double phi) {
    Path best; best.n = 0; best.total = 1e300;
    const double sp = std::sin(phi), cp = std::cos(phi);        // (sin, cos of -phi are -sp, cp: libm's sin is odd and its cos even to the last bit)
    const double xb = x * cp + y * sp, yb = x * sp - y * cp;    // the same word driven backwards
    const Pre q[4] = {make_pre(x, y, phi, sp, cp, true), make_pre(-x, y, -phi, -sp, cp, true), make_pre(x, -y, -phi, -sp, cp, true), make_pre(-x, -y, phi, sp, cp, true)};
    const Pre qb[4] = {make_pre(xb, yb, phi, sp, cp, false), make_pre(-xb, yb, -phi, -sp, cp, false), make_pre(xb, -yb, -phi, -sp, cp, false), make_pre(-xb, -yb, phi, sp, cp, false)};
    four(best, LpSpLp, q, "LSL", "RSR", 3, f_tuv);
    four(best, LpSpRp, q, "LSR", "RSL", 3, f_tuv);
    four(best, LpRmL, q, "LRL", "RLR", 3, f_tuv);
    four(best, LpRmL, qb, "LRL", "RLR", 3, f_vut);
    four(best, LpRupLumRm, q, "LRLR", "RLRL", 4, f_tuuv_m);
    four(best, LpRumLumRp, q, "LRLR", "RLRL", 4, f_tuuv_p);
    four(best, LpRmSmLm, q, "LRSL", "RLSR", 4, f_t_q_u_v);
    four(best, LpRmSmRm, q, "LRSR", "RLSL", 4, f_t_q_u_v);
    four(best, LpRmSmLm, qb, "LSRL", "RSLR", 4, f_v_u_q_t);
    four(best, LpRmSmRm, qb, "RSRL", "LSLR", 4, f_v_u_q_t);
    four(best, LpRmSLmRp, q, "LRSLR", "RLSRL", 5, f_t_q_u_q_v);
    return best;
}
// advance a pose along one segment by the signed amount s (unit radius)
static inline void advance(char ty, double s, double &x, double &y, double &yaw) {
    if (ty == 'S') { x += s * std::cos(yaw); y += s * std::sin(yaw); }
    else { const double k = ty == 'L' ? 1.0 : -1.0, y1 = yaw + k * s; x += k * (std::sin(y1) - std::sin(yaw)); y += -k * (std::cos(y1) - std::cos(yaw)); yaw = y1; }
}
// the same with sin / cos of the heading carried along (sy, cy: of `yaw` on entry and on exit) -- a walk along a path asks libm for each angle once
static inline void advance_sc(char ty, double s, double &x, double &y, double &yaw, double &sy, double &cy) {
    if (ty == 'S') { x += s * cy; y += s * sy; }
    else { const double k = ty == 'L' ? 1.0 : -1.0, y1 = yaw + k * s, s1 = std::sin(y1), c1 = std::cos(y1); x += k * (s1 - sy); y += -k * (c1 - cy); yaw = y1; sy = s1; cy = c1; }
}
}  // namespace rs

// ---------------------------------------------------------------- non-holonomic-WITH-obstacles heuristic on a lattice (round 6)
// Cost-to-go of the search's own motion model -- arcs of the same curvatures, the same reverse / steering / direction-switch costs -- on a coarse (x, y, yaw, direction of arrival)
// lattice, obstacles included (every cell-centre pose is collision-tested once), by ONE backward Dijkstra from the goal.  The reference's search has the two halves of this --
// holonomic WITH obstacles (a_star.jl distance map) and non-holonomic WITHOUT (Reeds-Shepp length, hybrid_a_star.jl:58) -- and takes their maximum; in a parking bay neither sees
// that the car must shunt, and the search pays with tens of thousands of expansions.  With this table the forward search expands little more than the nodes of its path.  The
// table depends on the obstacle field and the goal, not on the start: a BATCH of searches in one field whose goals lie within a metre of each other shares one table (each
// instance looks it up shifted by its goal's offset from the table's; the analytic expansion finishes the last metres exactly) -- obca_plan_hybrid_astar_batch2.
struct NhTable {
    double res = 0, yres = 0, xmin = 0, ymin = 0, gx = 0, gy = 0, gyaw = 0; int nx = 0, ny = 0, nyaw = 0;
    std::vector<float> h;      // [cell * 2 + (arrived driving forwards ? 0 : 1)]
    long long cells() const { return (long long)nx * ny * nyaw; }
    long long cell(double x, double y, double yaw) const {
        const long long ix = (long long)std::floor((x - xmin) / res), iy = (long long)std::floor((y - ymin) / res);
        if (ix < 0 || iy < 0 || ix >= nx || iy >= ny) return -1;
        long long ia = (long long)std::floor((wrap(yaw) + M_PI) / yres); if (ia >= nyaw) ia = nyaw - 1; if (ia < 0) ia = 0;
        return (iy * nx + ix) * nyaw + ia;
    }
    // cost-to-go of the pose for a car that arrived there driving in direction dir (0: start of a path: the cheaper of the two); < 0: not in the table
    double lookup(double x, double y, double yaw, int dir) const {
        const long long k = cell(x, y, yaw); if (k < 0) return -1.0;
        const float a = h[(size_t)k * 2], b_ = h[(size_t)k * 2 + 1], v = dir > 0 ? a : (dir < 0 ? b_ : std::min(a, b_));
        return v < 1e8f ? (double)v : -1.0;
    }
};
static void build_nh_table(const World &w, const double goal[3], double res, double yres, double smax, int nst, double L, double crev, double csw, double cst, double gtol, double ytol, NhTable &T) {
    T.res = res; T.yres = yres; T.xmin = w.xmin; T.ymin = w.ymin; T.gx = goal[0]; T.gy = goal[1]; T.gyaw = goal[2];
    T.nx = (int)std::ceil((w.xmax - w.xmin) / res) + 1; T.ny = (int)std::ceil((w.ymax - w.ymin) / res) + 1; T.nyaw = (int)std::ceil(2 * M_PI / yres);
    const long long nc = T.cells();
    T.h.assign((size_t)nc * 2, 1e9f);
    std::vector<uint8_t> free_((size_t)nc, 0);
    std::vector<double> cyaw(T.nyaw), syaw(T.nyaw);
    for (int ia = 0; ia < T.nyaw; ia++) { const double a = -M_PI + (ia + 0.5) * yres; cyaw[ia] = std::cos(a); syaw[ia] = std::sin(a); }
    for (int iy = 0; iy < T.ny; iy++) for (int ix = 0; ix < T.nx; ix++) for (int ia = 0; ia < T.nyaw; ia++)
        free_[(size_t)(((long long)iy * T.nx + ix) * T.nyaw + ia)] = !collides_cs(w, w.xmin + (ix + 0.5) * res, w.ymin + (iy + 0.5) * res, cyaw[ia], syaw[ia]);
    const double lstep = 1.5 * res;                                  // an arc long enough to leave its cell
    std::vector<double> kappa(2 * nst + 1), ecost(2 * nst + 1);
    for (int si = -nst; si <= nst; si++) { kappa[si + nst] = std::tan(smax * si / nst) / L; ecost[si + nst] = cst * std::fabs(smax * si / nst) * lstep; }
    typedef std::pair<float, long long> QE; std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
    // the goal: every free cell whose centre lies within the search's own goal tolerances
    for (int iy = 0; iy < T.ny; iy++) for (int ix = 0; ix < T.nx; ix++) {
        const double cx = w.xmin + (ix + 0.5) * res, cy = w.ymin + (iy + 0.5) * res;
        if (std::hypot(cx - goal[0], cy - goal[1]) > std::max(gtol, res)) continue;
        for (int ia = 0; ia < T.nyaw; ia++) {
            if (std::fabs(wrap(-M_PI + (ia + 0.5) * yres - goal[2])) > std::max(ytol, yres)) continue;
            const long long k = ((long long)iy * T.nx + ix) * T.nyaw + ia; if (!free_[(size_t)k]) continue;
            for (int di = 0; di < 2; di++) { T.h[(size_t)k * 2 + di] = 0.f; pq.push({0.f, k * 2 + di}); }
        }
    }
    while (!pq.empty()) {
        const QE e = pq.top(); pq.pop();
        if (e.first > T.h[(size_t)e.second]) continue;
        const long long k = e.second >> 1; const int di = (int)(e.second & 1), d = di ? -1 : 1;      // state: in cell k, having ARRIVED driving in direction d
        const int ia = (int)(k % T.nyaw); const long long cxy = k / T.nyaw; const int ix = (int)(cxy % T.nx), iy = (int)(cxy / T.nx);
        const double x1 = w.xmin + (ix + 0.5) * res, y1 = w.ymin + (iy + 0.5) * res, yaw1 = -M_PI + (ia + 0.5) * yres, s1 = syaw[ia], c1 = cyaw[ia];
        // predecessors: the poses from which the arc (d, steer) ends here = this pose driven along (-d, steer)
        for (int si = -nst; si <= nst; si++) {
            const double kap = kappa[si + nst], ds = -d * lstep; double x0, y0, yaw0;
            if (std::fabs(kap) < 1e-9) { x0 = x1 + ds * c1; y0 = y1 + ds * s1; yaw0 = yaw1; }
            else { yaw0 = yaw1 + ds * kap; x0 = x1 + (std::sin(yaw0) - s1) / kap; y0 = y1 - (std::cos(yaw0) - c1) / kap; }
            const long long kp = T.cell(x0, y0, yaw0);
            if (kp < 0 || kp == k || !free_[(size_t)kp]) continue;
            const float base = e.first + (float)(lstep * (d > 0 ? 1.0 : crev) + ecost[si + nst]);
            for (int dp = 0; dp < 2; dp++) {                         // the predecessor state: arrived at kp driving in direction (dp ? -1 : +1), then drives d
                const float c = base + (dp != di ? (float)csw : 0.f);
                if (c < T.h[(size_t)kp * 2 + dp]) { T.h[(size_t)kp * 2 + dp] = c; pq.push({c, kp * 2 + dp}); }
            }
        }
    }
}

extern "C" {

/*
 * Hybrid A* from start (x, y, yaw) to goal.  Obstacles as in obca_parking_signed_dist_batch (nOb, vOb = rows per obstacle, A row-major
 * (M x 2), b).  ego = [front, left, rear, right] extents from the rear axle (main.jl:73), L = wheelbase, XYbounds = [xmin,xmax,ymin,ymax].
 * opts (may be NULL) = {xy resolution 0.25, yaw resolution deg 7.5, primitive length 0.6, max steer 0.6, #steer samples per side 2,
 *                       collision margin 0.1, goal xy tolerance 0.3, goal yaw tolerance deg 8, reverse cost 1.5, switch cost 2.0,
 *                       steer cost 0.3, max expansions 400000, analytic (Reeds-Shepp) expansion 1, steer-change cost 0.2, heuristic weight 1,
 *                       Reeds-Shepp heuristic 0}  (the first nopts of these 16 doubles; the rest keep their defaults).
 * Output: path[3 * k] = x, y, yaw of the k-th node and dir[k] = +1 / -1 (motion that led to the node), up to cap nodes.
 * Returns the number of nodes (>= 2), 0 if no path was found, -1 on bad arguments, -2 if the start or the goal collides.
 */
static int hybrid_astar_impl(const double start[3], const double goal[3], int nOb, const int *vOb, const double *A, const double *b,
                             const double ego[4], double L, const double XYbounds[4], const double *opts_in, int nopts, double *path, int *dir, int cap,
                             int *expansions, const NhTable *shared) {
    if (!start || !goal || nOb < 0 || !vOb || !A || !b || !ego || !XYbounds || !path || !dir || cap < 2) return -1;
    if (opts_in && (nopts < 0 || nopts > OBCA_PLAN_NOPTS)) return -1;
    // the caller's first nopts options over the defaults: an array of the 14 options of rounds 1-4 leaves the newer ones at their defaults and nothing is read beyond its end
    double optbuf[OBCA_PLAN_NOPTS] = {0.25, 7.5, 0.6, 0.6, 2, 0.1, 0.3, 8.0, 1.5, 2.0, 0.3, 400000, 1.0, 0.2, 1.0, 0.0, 0.0, 7.5};
    for (int i = 0; opts_in && i < nopts; i++) optbuf[i] = opts_in[i];
    const double *opts = optbuf;
    if (!(opts[14] > 0.0) || !std::isfinite(opts[14])) return -1;      // heuristic weight: positive and finite (0, negative or NaN would silently corrupt the search order)
    const double res = opts ? opts[0] : 0.25, yres = (opts ? opts[1] : 7.5) * M_PI / 180, step = opts ? opts[2] : 0.6, smax = opts ? opts[3] : 0.6;
    const int nst = opts ? (int)opts[4] : 2; const double margin = opts ? opts[5] : 0.1, gtol = opts ? opts[6] : 0.3, ytol = (opts ? opts[7] : 8.0) * M_PI / 180;
    const double crev = opts ? opts[8] : 1.5, csw = opts ? opts[9] : 2.0, cst = opts ? opts[10] : 0.3; const long maxexp = opts ? (long)opts[11] : 400000;
    const double analytic = opts ? opts[12] : 1.0;
    const double hweight = opts ? opts[14] : 1.0;       // weight of the heuristic (hybrid_a_star.jl:64 H_COST; > 1: greedier search, fewer expansions, longer paths)
    const bool rs_heur = opts ? opts[15] != 0.0 : false; // max(grid heuristic, Reeds-Shepp length) as the heuristic (hybrid_a_star.jl:58 USE_NONHOLONOMIC_WITHOUT_OBSTACLE_HEURISTIC)
    const double nh_res = opts[16], nh_yres = opts[17] * M_PI / 180;      // lattice of the non-holonomic-with-obstacles heuristic (xy cell [m], 0 = off; yaw cell): build_nh_table
    if (nh_res < 0 || !std::isfinite(nh_res) || (nh_res > 0 && (nh_res < 0.05 || !(nh_yres > 1e-3)))) return -1;
    const double cchg = opts ? opts[13] : 0.2;          // steer-change cost per radian (hybrid_a_star.jl:63 STEER_CHANGE_COST)       // analytic (Reeds-Shepp) expansion towards the goal, hybrid_a_star.jl:193-214: 0 = off,
                                                         // else the fraction of the steering lock its arcs use (1 = the reference's full lock)
    World w; w.nOb = nOb; w.v.assign(vOb, vOb + nOb); w.off.assign(nOb + 1, 0);
    for (int j = 0; j < nOb; j++) { if (vOb[j] < 1) return -1; w.off[j + 1] = w.off[j] + vOb[j]; }
    w.A.assign(A, A + 2 * w.off[nOb]); w.b.assign(b, b + w.off[nOb]);
    w.xmin = XYbounds[0]; w.xmax = XYbounds[1]; w.ymin = XYbounds[2]; w.ymax = XYbounds[3];
    std::memcpy(w.ego, ego, sizeof w.ego); w.margin = margin; w.finish();
    if (collides(w, start[0], start[1], start[2]) || collides(w, goal[0], goal[1], goal[2])) return -2;

    // holonomic heuristic: Dijkstra on the xy grid from the goal for a disc of the car's half width (hybrid_a_star.jl's grid heuristic)
    const int nx = (int)std::ceil((w.xmax - w.xmin) / res) + 1, ny = (int)std::ceil((w.ymax - w.ymin) / res) + 1;
    std::vector<float> hmap((size_t)nx * ny, 1e9f);
    {
        const double rad = std::min(std::min(ego[1], ego[3]), 0.5 * (ego[0] + ego[2])) * 0.9;
        std::vector<uint8_t> blocked((size_t)nx * ny, 0);
        for (int iy = 0; iy < ny; iy++) for (int ix = 0; ix < nx; ix++) blocked[(size_t)iy * nx + ix] = disc_collides(w, w.xmin + ix * res, w.ymin + iy * res, rad);
        typedef std::pair<float, int> QE; std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
        const int gx = std::min(nx - 1, std::max(0, (int)std::lround((goal[0] - w.xmin) / res))), gy = std::min(ny - 1, std::max(0, (int)std::lround((goal[1] - w.ymin) / res)));
        hmap[(size_t)gy * nx + gx] = 0; pq.push({0.f, gy * nx + gx});
        while (!pq.empty()) {
            QE e = pq.top(); pq.pop();
            if (e.first > hmap[e.second]) continue;
            const int cx = e.second % nx, cy = e.second / nx;
            for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy) continue;
                const int qx = cx + dx, qy = cy + dy; if (qx < 0 || qy < 0 || qx >= nx || qy >= ny) continue;
                const size_t q = (size_t)qy * nx + qx; if (blocked[q]) continue;
                const float c = e.first + (float)(res * std::hypot(dx, dy));
                if (c < hmap[q]) { hmap[q] = c; pq.push({c, (int)q}); }
            }
        }
    }
    // the lattice heuristic: the batch's shared table (looked up shifted by this goal's offset from the table's goal) or one built for this search
    NhTable own; const NhTable *nh = nullptr; double nh_dx = 0, nh_dy = 0;
    if (nh_res > 0) {
        if (shared && shared->res == nh_res && std::hypot(goal[0] - shared->gx, goal[1] - shared->gy) <= 1.5 && std::fabs(wrap(goal[2] - shared->gyaw)) <= 1e-9) { nh = shared; nh_dx = goal[0] - shared->gx; nh_dy = goal[1] - shared->gy; }
        else { build_nh_table(w, goal, nh_res, nh_yres, smax, nst, L, crev, csw, cst, gtol, ytol, own); nh = &own; }
    }
    auto heur = [&](double x, double y, double yaw, int dir_in) {
        const int ix = std::min(nx - 1, std::max(0, (int)std::lround((x - w.xmin) / res))), iy = std::min(ny - 1, std::max(0, (int)std::lround((y - w.ymin) / res)));
        const double h2 = hmap[(size_t)iy * nx + ix] < 1e8f ? hmap[(size_t)iy * nx + ix] : std::hypot(x - goal[0], y - goal[1]) + 5.0;
        const double Rmin = L / std::tan(smax);
        double h = std::max(h2, Rmin * std::fabs(wrap(yaw - goal[2])) * 0.5);
        if (rs_heur) {      // length of the shortest Reeds-Shepp curve to the goal: what the car needs at least, obstacles ignored (hybrid_a_star.jl:58, 542: max(c_h_dp, c_h_rs))
            const double dx = goal[0] - x, dy = goal[1] - y, c = std::cos(yaw), s_ = std::sin(yaw);
            const rs::Path p = rs::shortest((c * dx + s_ * dy) / Rmin, (-s_ * dx + c * dy) / Rmin, wrap(goal[2] - yaw));
            if (p.n > 0) h = std::max(h, Rmin * p.total);
        }
        if (nh) { const double hn = nh->lookup(x - nh_dx, y - nh_dy, yaw, dir_in); if (hn >= 0) h = std::max(h, hn); }
        return hweight * h;
    };
    const int nyaw = (int)std::ceil(2 * M_PI / yres);
    auto key = [&](double x, double y, double yaw) -> long long {
        const long long ix = (long long)std::floor((x - w.xmin) / res), iy = (long long)std::floor((y - w.ymin) / res);
        long long ia = (long long)std::floor((wrap(yaw) + M_PI) / yres); if (ia >= nyaw) ia = nyaw - 1;
        return (iy * (nx + 1) + ix) * nyaw + ia;
    };
    std::vector<Node> nodes; nodes.reserve(1 << 16);
    // cheapest cost-to-come seen per (x, y, yaw) cell: a dense table that lives with the thread (a search touches a small part of it: those cells are reset afterwards)
    static thread_local std::vector<double> best; static thread_local std::vector<long long> touched;
    const long long ncell = (long long)(nx + 1) * (ny + 1) * nyaw;
    if ((long long)best.size() < ncell) best.assign((size_t)ncell, 1e300);
    struct Reset { std::vector<double> &b; std::vector<long long> &t; ~Reset() { for (long long k : t) b[(size_t)k] = 1e300; t.clear(); } } reset_{best, touched};
    auto cell = [&](double x, double y, double yaw) -> long long { const long long k = key(x, y, yaw); return k >= 0 && k < ncell ? k : -1; };
    typedef std::pair<double, int> QE; std::priority_queue<QE, std::vector<QE>, std::greater<QE>> open;
    nodes.push_back({start[0], start[1], wrap(start[2]), 0.0, -1, 0, 0});
    open.push({heur(start[0], start[1], start[2], 0), 0}); { const long long k = cell(start[0], start[1], start[2]); if (k >= 0) { best[(size_t)k] = 0.0; touched.push_back(k); } }
    long nexp = 0; int found = -1;
    const int sub = std::max(1, (int)std::ceil(step / 0.2));
    const double Rmin = L / std::tan(smax * (analytic > 0 ? std::min(1.0, analytic) : 1.0));
    std::vector<double> kappa(2 * nst + 1); for (int si = -nst; si <= nst; si++) kappa[si + nst] = std::tan(smax * si / nst) / L;      // curvature of every steering command
    std::vector<double> tail; std::vector<int> taild;     // collision-free Reeds-Shepp connection of node `found` to the exact goal pose
    while (!open.empty() && nexp < maxexp) {
        const QE e = open.top(); open.pop();
        const Node cur = nodes[e.second];
        { const long long k = cell(cur.x, cur.y, cur.yaw); if (k >= 0 && best[(size_t)k] < cur.g - 1e-9) continue; }
        const double dgoal = std::hypot(cur.x - goal[0], cur.y - goal[1]);
        if (dgoal <= gtol && std::fabs(wrap(cur.yaw - goal[2])) <= ytol) { found = e.second; break; }
        nexp++;
        const double sy0 = std::sin(cur.yaw), cy0 = std::cos(cur.yaw);
        if (analytic > 0 && (dgoal < 6.0 * Rmin || nexp % 16 == 0)) {
            // shortest Reeds-Shepp curve from this node to the goal: if the car can follow it without touching anything, the search is over
            const double dx = goal[0] - cur.x, dy = goal[1] - cur.y, c = cy0, s_ = sy0;
            const rs::Path p = rs::shortest((c * dx + s_ * dy) / Rmin, (-s_ * dx + c * dy) / Rmin, wrap(goal[2] - cur.yaw));
            // the walk along it, sampled every <= 0.2 m: nearly every curve hits something within a few samples, so the first pass only tests (heading cos / sin by the
            // addition theorem from those the walk carries) and the one curve that passes is walked again to record its poses
            auto walk = [&](bool record) {
                double x = 0, y = 0, yaw = 0, sy = std::sin(0.0), cy = std::cos(0.0);
                for (int i = 0; i < p.n; i++) {
                    const double L_ = std::fabs(p.len[i]); if (L_ < 1e-12) continue;
                    const int d = p.len[i] >= 0 ? 1 : -1, m = std::max(1, (int)std::ceil(L_ * Rmin / 0.2));
                    for (int q = 0; q < m; q++) {
                        rs::advance_sc(p.type[i], p.len[i] / m, x, y, yaw, sy, cy);
                        const double wx = cur.x + Rmin * (c * x - s_ * y), wy = cur.y + Rmin * (s_ * x + c * y);
                        if (record) { tail.insert(tail.end(), {wx, wy, wrap(cur.yaw + yaw)}); taild.push_back(d); }
                        else if (collides_cs(w, wx, wy, c * cy - s_ * sy, s_ * cy + c * sy)) return false;
                    }
                }
                return true;
            };
            if (p.n > 0 && walk(false)) { tail.clear(); taild.clear(); walk(true); found = e.second; break; }
        }
        for (int d = 1; d >= -1; d -= 2)
            for (int si = -nst; si <= nst; si++) {
                const double steer = smax * si / nst, kap = kappa[si + nst];
                double x = cur.x, y = cur.y, yaw = cur.yaw, sy = sy0, cy = cy0; bool ok = true;      // (sy, cy: sin, cos of yaw -- each angle goes to libm once)
                for (int q = 0; q < sub && ok; q++) {
                    const double ds = d * step / sub;
                    if (std::fabs(kap) < 1e-9) { x += ds * cy; y += ds * sy; }
                    else { const double y1 = yaw + ds * kap, s1 = std::sin(y1), c1 = std::cos(y1); x += (s1 - sy) / kap; y += -(c1 - cy) / kap; yaw = y1; sy = s1; cy = c1; }
                    ok = !collides_cs(w, x, y, cy, sy);
                }
                if (!ok) continue;
                double g = cur.g + step * (d > 0 ? 1.0 : crev) + cst * std::fabs(steer) * step + cchg * std::fabs(steer - smax * cur.steer / nst);
                if (cur.dir != 0 && cur.dir != d) g += csw;
                const long long k = cell(x, y, yaw);
                if (k < 0) continue;                                   // (cannot happen inside XYbounds)
                if (best[(size_t)k] <= g) continue;
                if (best[(size_t)k] == 1e300) touched.push_back(k);
                best[(size_t)k] = g;
                nodes.push_back({x, y, wrap(yaw), g, e.second, (int8_t)d, (int8_t)si});
                open.push({g + heur(x, y, yaw, d), (int)nodes.size() - 1});
            }
    }
    if (expansions) *expansions = (int)nexp;
    if (found < 0) return 0;
    std::vector<int> chain; for (int i = found; i >= 0; i = nodes[i].parent) chain.push_back(i);
    std::reverse(chain.begin(), chain.end());
    // densify: replay each primitive at 0.2 m so that the caller can resample by arc length
    std::vector<double> px; std::vector<int> pd;
    px.insert(px.end(), {nodes[chain[0]].x, nodes[chain[0]].y, nodes[chain[0]].yaw}); pd.push_back(nodes[chain.size() > 1 ? chain[1] : chain[0]].dir);
    for (size_t c = 1; c < chain.size(); c++) {
        const Node &p = nodes[chain[c - 1]], &n = nodes[chain[c]];
        const double steer = smax * n.steer / nst, kap = std::tan(steer) / L;
        double x = p.x, y = p.y, yaw = p.yaw;
        for (int q = 0; q < sub; q++) {
            const double ds = n.dir * step / sub;
            if (std::fabs(kap) < 1e-9) { x += ds * std::cos(yaw); y += ds * std::sin(yaw); }
            else { const double y1 = yaw + ds * kap; x += (std::sin(y1) - std::sin(yaw)) / kap; y += -(std::cos(y1) - std::cos(yaw)) / kap; yaw = y1; }
            px.insert(px.end(), {x, y, yaw}); pd.push_back(n.dir);
        }
    }
    px.insert(px.end(), tail.begin(), tail.end()); pd.insert(pd.end(), taild.begin(), taild.end());
    if (chain.size() == 1 && !taild.empty()) pd[0] = taild[0];
    const int cnt = (int)pd.size();
    if (cnt > cap) return -1;
    std::memcpy(path, px.data(), sizeof(double) * 3 * cnt); std::memcpy(dir, pd.data(), sizeof(int) * cnt);
    return cnt;
}

/* Shortest Reeds-Shepp path from start to goal (x, y, yaw) for turning radius R, sampled every `step` metres: path[3k..] = pose k,
 * dir[k] = +1 / -1.  word (>= 6 chars, may be NULL) receives the segment types ("LSR", "LRSLR", ...), seglen (5, may be NULL) their signed
 * lengths in metres.  Returns the number of samples (start and goal included), -1 on bad arguments / cap too small. */
int obca_plan_hybrid_astar2(const double start[3], const double goal[3], int nOb, const int *vOb, const double *A, const double *b,
                            const double ego[4], double L, const double XYbounds[4], const double *opts, int nopts, double *path, int *dir, int cap,
                            int *expansions) {
    return hybrid_astar_impl(start, goal, nOb, vOb, A, b, ego, L, XYbounds, opts, nopts, path, dir, cap, expansions, nullptr);
}

int obca_plan_hybrid_astar_batch2(int B, const double *starts, const double *goals, int nOb, const int *vOb, const double *A, const double *b,
                                  const double ego[4], double L, const double XYbounds[4], const double *opts, int nopts, double *paths, int *dirs, int cap,
                                  int *counts, int *expansions, int threads) {
    if (B < 0 || !starts || !goals || !paths || !dirs || !counts || cap < 2) return -1;
    if (opts && (nopts < 0 || nopts > OBCA_PLAN_NOPTS)) return -1;
    // the lattice heuristic (options 16, 17) is a property of the obstacle field and the goal: ONE table for the batch, built around the goal in the middle of the batch's goals
    NhTable table; const NhTable *shared = nullptr;
    if (B > 0 && opts && nopts > 16 && opts[16] > 0 && nOb >= 0 && vOb && A && b && ego && XYbounds) {
        double o_[OBCA_PLAN_NOPTS] = {0.25, 7.5, 0.6, 0.6, 2, 0.1, 0.3, 8.0, 1.5, 2.0, 0.3, 400000, 1.0, 0.2, 1.0, 0.0, 0.0, 7.5};
        for (int i = 0; i < nopts; i++) o_[i] = opts[i];
        if (std::isfinite(o_[16]) && o_[16] >= 0.05 && o_[17] > 0.05 && (int)o_[4] >= 0) {
            std::vector<double> gxs(B), gys(B);
            for (int i = 0; i < B; i++) { gxs[i] = goals[3 * (size_t)i]; gys[i] = goals[3 * (size_t)i + 1]; }
            std::nth_element(gxs.begin(), gxs.begin() + B / 2, gxs.end()); std::nth_element(gys.begin(), gys.begin() + B / 2, gys.end());
            const double gref[3] = {gxs[B / 2], gys[B / 2], goals[2]};
            World w; w.nOb = nOb; w.v.assign(vOb, vOb + nOb); w.off.assign(nOb + 1, 0); bool ok_ = true;
            for (int j = 0; j < nOb; j++) { if (vOb[j] < 1) ok_ = false; w.off[j + 1] = w.off[j] + vOb[j]; }
            if (ok_) {
                w.A.assign(A, A + 2 * w.off[nOb]); w.b.assign(b, b + w.off[nOb]);
                w.xmin = XYbounds[0]; w.xmax = XYbounds[1]; w.ymin = XYbounds[2]; w.ymax = XYbounds[3];
                std::memcpy(w.ego, ego, sizeof w.ego); w.margin = o_[5]; w.finish();
                build_nh_table(w, gref, o_[16], o_[17] * M_PI / 180, o_[3], (int)o_[4], L, o_[8], o_[9], o_[10], o_[6], o_[7] * M_PI / 180, table);
                shared = &table;
            }
        }
    }
    unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > (unsigned)B) nt = (unsigned)(B > 0 ? B : 1);
    std::atomic<int> next(0);      // the searches differ by two orders of magnitude in length: a shared counter, not static slices
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < B; i = next.fetch_add(1)) {
            int ne = 0;
            counts[i] = hybrid_astar_impl(starts + 3 * (size_t)i, goals + 3 * (size_t)i, nOb, vOb, A, b, ego, L, XYbounds, opts, nopts, paths + 3 * (size_t)cap * i,
                                          dirs + (size_t)cap * i, cap, &ne, shared);
            if (expansions) expansions[i] = ne;
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return 0;
}

// The entry points of rounds 1-4 took an option array WITHOUT a length, documented as 14 doubles; round 5 read two more (heuristic weight, Reeds-Shepp heuristic) from the same
// pointer -- an out-of-bounds read for every caller built against the old header.  They keep the 14-double meaning (weight 1, no Reeds-Shepp heuristic); the *2 forms carry the length.
int obca_plan_hybrid_astar(const double start[3], const double goal[3], int nOb, const int *vOb, const double *A, const double *b,
                           const double ego[4], double L, const double XYbounds[4], const double *opts, double *path, int *dir, int cap,
                           int *expansions) {
    return obca_plan_hybrid_astar2(start, goal, nOb, vOb, A, b, ego, L, XYbounds, opts, 14, path, dir, cap, expansions);
}
int obca_plan_hybrid_astar_batch(int B, const double *starts, const double *goals, int nOb, const int *vOb, const double *A, const double *b,
                                 const double ego[4], double L, const double XYbounds[4], const double *opts, double *paths, int *dirs, int cap,
                                 int *counts, int *expansions, int threads) {
    return obca_plan_hybrid_astar_batch2(B, starts, goals, nOb, vOb, A, b, ego, L, XYbounds, opts, 14, paths, dirs, cap, counts, expansions, threads);
}

int obca_plan_reeds_shepp(const double start[3], const double goal[3], double R, double step, double *path, int *dir, int cap, char *word,
                          double *seglen, double *total) {
    if (!start || !goal || !(R > 0) || !(step > 0) || !path || !dir || cap < 2) return -1;
    const double dx = goal[0] - start[0], dy = goal[1] - start[1], c = std::cos(start[2]), s_ = std::sin(start[2]);
    const rs::Path p = rs::shortest((c * dx + s_ * dy) / R, (-s_ * dx + c * dy) / R, wrap(goal[2] - start[2]));
    if (p.n == 0) return -1;
    if (word) { for (int i = 0; i < p.n; i++) word[i] = p.type[i]; word[p.n] = 0; }
    if (seglen) for (int i = 0; i < 5; i++) seglen[i] = i < p.n ? p.len[i] * R : 0.0;
    if (total) *total = p.total * R;
    int cnt = 0;
    double x = 0, y = 0, yaw = 0;                                    // normalised frame
    auto emit = [&](int d) -> bool {
        if (cnt >= cap) return false;
        path[3 * cnt] = start[0] + R * (c * x - s_ * y); path[3 * cnt + 1] = start[1] + R * (s_ * x + c * y); path[3 * cnt + 2] = wrap(start[2] + yaw);
        dir[cnt++] = d; return true;
    };
    if (!emit(p.len[0] >= 0 ? 1 : -1)) return -1;
    for (int i = 0; i < p.n; i++) {
        const double L_ = std::fabs(p.len[i]); if (L_ < 1e-12) continue;
        const int d = p.len[i] >= 0 ? 1 : -1, m = std::max(1, (int)std::ceil(L_ * R / step));
        for (int q = 0; q < m; q++) { rs::advance(p.type[i], p.len[i] / m, x, y, yaw); if (!emit(d)) return -1; }
    }
    return cnt;
}

/* 1 if the car pose collides with an obstacle (inflated by margin) or leaves XYbounds -- the planner's own test, exported for the tests */
int obca_plan_collides(double x, double y, double yaw, int nOb, const int *vOb, const double *A, const double *b, const double ego[4],
                       const double XYbounds[4], double margin) {
    if (nOb < 0 || (nOb > 0 && (!vOb || !A || !b)) || !ego || !XYbounds) return -1;
    for (int j = 0; j < nOb; j++) if (vOb[j] < 1) return -1;
    World w; w.nOb = nOb; w.v.assign(vOb, vOb + nOb); w.off.assign(nOb + 1, 0);
    for (int j = 0; j < nOb; j++) w.off[j + 1] = w.off[j] + vOb[j];
    w.A.assign(A, A + 2 * w.off[nOb]); w.b.assign(b, b + w.off[nOb]);
    w.xmin = XYbounds[0]; w.xmax = XYbounds[1]; w.ymin = XYbounds[2]; w.ymax = XYbounds[3];
    std::memcpy(w.ego, ego, sizeof w.ego); w.margin = margin; w.finish();
    return collides(w, x, y, yaw) ? 1 : 0;
}

/*
 * 3-D grid A* for the quadcopter warm start (the role of a_star_3D.jl / mainQuadcopter.jl:108-138): 26-connected grid of spacing `res`
 * over the room [0, room[0]] x [0, room[1]] x [0, room[2]], boxes = nBox x 6 as [xmax,ymax,zmax,-xmin,-ymin,-zmin] inflated by `clear`
 * (ball radius + margin), Euclidean heuristic.  Output: path[3k..] way-points from start to goal (grid nodes, end points exact),
 * at most cap.  Returns the number of way-points (>= 2), 0 = no path, -1 = bad arguments, -2 = start or goal inside an inflated box.
 */
int obca_plan_astar3d(const double start[3], const double goal[3], int nBox, const double *boxes, double clear, const double room[3],
                      double res, double *path, int cap, int *expansions) {
    if (!start || !goal || nBox < 0 || (nBox && !boxes) || !room || !path || cap < 2 || !(res > 0)) return -1;
    const int nx = (int)std::floor(room[0] / res) + 1, ny = (int)std::floor(room[1] / res) + 1, nz = (int)std::floor(room[2] / res) + 1;
    if ((long long)nx * ny * nz > 64000000LL) return -1;
    auto blocked = [&](double x, double y, double z) {
        if (x < 0 || y < 0 || z < 0 || x > room[0] || y > room[1] || z > room[2]) return true;
        for (int j = 0; j < nBox; j++) {
            const double *b = boxes + 6 * j;
            if (x <= b[0] + clear && y <= b[1] + clear && z <= b[2] + clear && x >= -b[3] - clear && y >= -b[4] - clear && z >= -b[5] - clear) return true;
        }
        return false;
    };
    if (blocked(start[0], start[1], start[2]) || blocked(goal[0], goal[1], goal[2])) return -2;
    auto cell = [&](const double p[3], int c[3]) { for (int i = 0; i < 3; i++) c[i] = (int)std::lround(p[i] / res); c[0] = std::min(c[0], nx - 1); c[1] = std::min(c[1], ny - 1); c[2] = std::min(c[2], nz - 1); };
    int cs[3], cg[3]; cell(start, cs); cell(goal, cg);
    auto id = [&](int x, int y, int z) { return ((size_t)z * ny + y) * nx + x; };
    const size_t ncell = (size_t)nx * ny * nz;
    std::vector<float> g(ncell, 1e30f); std::vector<int> par(ncell, -1);
    typedef std::pair<float, int> QE; std::priority_queue<QE, std::vector<QE>, std::greater<QE>> open;
    auto h = [&](int x, int y, int z) { return (float)(res * std::sqrt((double)(x - cg[0]) * (x - cg[0]) + (double)(y - cg[1]) * (y - cg[1]) + (double)(z - cg[2]) * (z - cg[2]))); };
    const size_t s0 = id(cs[0], cs[1], cs[2]), gid = id(cg[0], cg[1], cg[2]);
    g[s0] = 0; open.push({h(cs[0], cs[1], cs[2]), (int)s0});
    long nexp = 0; bool found = false;
    while (!open.empty()) {
        const QE e = open.top(); open.pop();
        const size_t c = (size_t)e.second; const int cx = (int)(c % nx), cy = (int)((c / nx) % ny), cz = (int)(c / ((size_t)nx * ny));
        if (e.first > g[c] + h(cx, cy, cz) + 1e-4f) continue;
        if (c == gid) { found = true; break; }
        nexp++;
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy && !dz) continue;
            const int qx = cx + dx, qy = cy + dy, qz = cz + dz;
            if (qx < 0 || qy < 0 || qz < 0 || qx >= nx || qy >= ny || qz >= nz) continue;
            if (blocked(qx * res, qy * res, qz * res)) continue;
            const size_t q = id(qx, qy, qz); const float gn = g[c] + (float)(res * std::sqrt((double)(dx * dx + dy * dy + dz * dz)));
            if (gn < g[q]) { g[q] = gn; par[q] = (int)c; open.push({gn + h(qx, qy, qz), (int)q}); }
        }
    }
    if (expansions) *expansions = (int)nexp;
    if (!found) return 0;
    std::vector<size_t> chain; for (long c = (long)gid; c >= 0; c = par[(size_t)c]) chain.push_back((size_t)c);
    std::reverse(chain.begin(), chain.end());
    const int cnt = (int)chain.size() + 2;
    if (cnt > cap) return -1;
    int k = 0; path[0] = start[0]; path[1] = start[1]; path[2] = start[2]; k = 1;
    for (size_t c : chain) { path[3 * k] = (c % nx) * res; path[3 * k + 1] = ((c / nx) % ny) * res; path[3 * k + 2] = (c / ((size_t)nx * ny)) * res; k++; }
    path[3 * k] = goal[0]; path[3 * k + 1] = goal[1]; path[3 * k + 2] = goal[2]; k++;
    return k;
}

}  // extern "C"
