/*
 * obca_plan.h -- C ABI of libobca_plan.so: host-side (CPU) warm-start planner for the parking scenarios.
 *
 * Replaces, for the purposes of producing the warm start the signed-distance NLP needs, the planner the reference calls before the
 * hot path:  hybrid_a_star(...) at AutonomousParking/main.jl:216-236 (hybrid_a_star.jl, with collision_check.jl / reeds_shepp.jl /
 * a_star.jl behind it).  It is NOT on the GPU hot path; it is the step before it ("next-2").
 * Obstacles use the same convention as include/obca_hip.h: nOb convex obstacles, vOb[j] half-space rows each, A row-major (M x 2), b.
 */
#ifndef OBCA_PLAN_H
#define OBCA_PLAN_H
#ifdef __cplusplus
extern "C" {
#endif

#define OBCA_PLAN_NOPTS 18      /* options the search knows (below) */

/* Hybrid A* from start (x, y, yaw) to goal.  ego = [front, left, rear, right] extents from the rear axle (main.jl:73), L = wheelbase,
 * XYbounds = [xmin, xmax, ymin, ymax] of the rear-axle position.
 * opts (NULL = defaults), nopts = how many doubles the caller's array holds (0 .. OBCA_PLAN_NOPTS; options beyond it keep their defaults, nothing beyond it is read):
 *   {xy resolution 0.25, yaw resolution [deg] 7.5, primitive length 0.6, max steer 0.6, steer samples per side 2,
 *   collision margin 0.1, goal xy tolerance 0.3, goal yaw tolerance [deg] 8, reverse cost 1.5, switch cost 2.0, steer cost 0.3,
 *   max expansions 400000, analytic expansion (1: try the shortest Reeds-Shepp curve to the goal from expanded nodes, as the reference does) 1,
 *   steer-change cost per radian 0.2, weight of the heuristic 1 (hybrid_a_star.jl:64 H_COST; must be positive and finite),
 *   Reeds-Shepp length as a second heuristic 0 (hybrid_a_star.jl:58),
 *   lattice heuristic: xy cell [m] 0 (0 = off) and yaw cell [deg] 7.5 of a NON-HOLONOMIC-WITH-OBSTACLES cost-to-go table -- the search's own arcs and costs on a coarse
 *   (x, y, yaw, direction of arrival) lattice, every cell pose collision-tested, one backward Dijkstra from the goal; the heuristic is the maximum of it and the others.  Where the
 *   reference's two heuristics (holonomic with obstacles, non-holonomic without) do not see that the car must shunt, the search expands 10-100 x fewer nodes with it.  The
 *   table takes 0.1-2 s to build: the batch call builds ONE for all its searches (goals within 1.5 m of the batch's median goal, equal goal headings; the others build their own)}.
 * Output: path[3k..3k+2] = x, y, yaw of node k (0.2 m apart), dir[k] = +1 / -1 (motion that led to the node), at most cap nodes.
 * Returns the number of nodes (>= 2); 0 = no path; -1 = bad arguments (nopts out of range, heuristic weight <= 0 or not finite) / cap too small; -2 = start or goal pose collides. */
int obca_plan_hybrid_astar2(const double start[3], const double goal[3], int nOb, const int *vOb, const double *A, const double *b,
                            const double ego[4], double L, const double XYbounds[4], const double *opts, int nopts, double *path, int *dir, int cap,
                            int *expansions /* may be NULL */);
/* The entry point of rounds 1-4: opts (NULL = defaults) is an array of exactly the FIRST 14 options above; heuristic weight 1, no Reeds-Shepp heuristic. */
int obca_plan_hybrid_astar(const double start[3], const double goal[3], int nOb, const int *vOb, const double *A, const double *b,
                           const double ego[4], double L, const double XYbounds[4], const double *opts, double *path, int *dir, int cap,
                           int *expansions /* may be NULL */);

/* The same search for B independent (start, goal) pairs in one obstacle field, on `threads` host threads (0 = one per hardware thread) inside the library: what a rank
 * calls for its slice of a batch before the solve (main.jl:216-252 plans one instance; a batch of 2 048 parallel-parking starts takes ~0.2 core-seconds each).
 * starts / goals: B x 3; paths: B x cap x 3; dirs: B x cap; counts[i] = what obca_plan_hybrid_astar2 returns for pair i; expansions (may be NULL): B.
 * Returns 0, or -1 on bad arguments. */
int obca_plan_hybrid_astar_batch2(int B, const double *starts, const double *goals, int nOb, const int *vOb, const double *A, const double *b,
                                  const double ego[4], double L, const double XYbounds[4], const double *opts, int nopts, double *paths, int *dirs, int cap,
                                  int *counts, int *expansions /* may be NULL */, int threads);
/* rounds 1-4 form: opts = the first 14 options */
int obca_plan_hybrid_astar_batch(int B, const double *starts, const double *goals, int nOb, const int *vOb, const double *A, const double *b,
                                 const double ego[4], double L, const double XYbounds[4], const double *opts, double *paths, int *dirs, int cap,
                                 int *counts, int *expansions /* may be NULL */, int threads);

/* REFERENCE mode: the reference's own Hybrid A* restated step by step (obca_amd/csrc/obca_planner_ref.cpp follows hybrid_a_star.jl:104-552, a_star.jl:47-281,
 * collision_check.jl:31-98): what main.jl:216-219 calls.  Obstacles are the POINT CLOUD main.jl builds (ox, oy: nob points), not H-representations.
 * opts (may be NULL = the reference's constants) = {xy grid 0.3, yaw grid deg 5, motion step 0.1, steer commands per side 5, max steer 0.6, wheelbase 2.7,
 *   switch-back cost 10, reverse factor 0, steer-change cost 10, steer cost 0, heuristic weight 1, disc radius of the heuristic 1.0, max expansions 2e6}.
 * Output: path[3k..3k+2] = x, y, yaw of the k-th pose (0.1 m apart; rx, ry, ryaw of the reference), at most cap poses.
 * Returns the number of poses (>= 2); 0 = no path; -1 = bad arguments / cap too small. */
int obca_plan_reference_hybrid_astar(const double start[3], const double goal[3], int nob, const double *ox, const double *oy, const double *opts,
                                     double *path, int cap, int *expansions /* may be NULL */);

/* Shortest Reeds-Shepp path (forward and reverse arcs of radius R and straight lines; stands where hybrid_a_star.jl:262-300 calls
 * reeds_shepp.calc_shortest_path, reeds_shepp.jl) from start to goal (x, y, yaw), sampled every `step` metres: path[3k..3k+2] = pose k,
 * dir[k] = +1 / -1.  word (>= 6 chars, may be NULL) receives the segment types ("LSR", "LRSLR", ...), seglen (5 doubles, may be NULL) their
 * signed lengths in metres, total (may be NULL) the path length.  Returns the number of samples, -1 on bad arguments / cap too small. */
int obca_plan_reeds_shepp(const double start[3], const double goal[3], double R, double step, double *path, int *dir, int cap, char *word,
                          double *seglen, double *total);

/* 1 if the car rectangle at (x, y, yaw), inflated by margin, overlaps an obstacle or leaves XYbounds (the planner's own test). */
int obca_plan_collides(double x, double y, double yaw, int nOb, const int *vOb, const double *A, const double *b, const double ego[4],
                       const double XYbounds[4], double margin);

/* 3-D grid A* for the quadcopter warm start (stands where mainQuadcopter.jl:124-128 calls a_star.calc_astar_path, a_star_3D.jl): 26-connected
 * grid of spacing res over the room [0,room[0]] x [0,room[1]] x [0,room[2]]; boxes nBox x 6 as [xmax,ymax,zmax,-xmin,-ymin,-zmin], inflated by
 * `clear`.  path receives way-points start .. goal.  Returns their number, 0 = no path, -1 = bad arguments, -2 = start / goal blocked. */
int obca_plan_astar3d(const double start[3], const double goal[3], int nBox, const double *boxes, double clear, const double room[3],
                      double res, double *path, int cap, int *expansions /* may be NULL */);

/* REFERENCE mode of the quadcopter's path search: QuadcopterNavigation/a_star_3D.jl restated (calc_astar_path :49-154, the 26-neighbour motion model :157-187,
 * calc_obstacle_map :193-231, get_final_path :233-263), as mainQuadcopter.jl:116-121 calls it.  All lengths in GRID UNITS of `reso` as in the reference (its caller scales
 * the room by 10 and passes reso = 1.0).  ox, oy, oz: the obstacle POINT lists (nob points; the reference appends the two room corners to them, :196-198 -- done inside).
 * A cell is blocked when its nearest obstacle point is within VEHICLE_RADIUS / reso = 2.5 / reso (:31, :221-224); the search is A* with the heuristic 1.1 x the Euclidean
 * distance (H_WEIGHT, :32), node keys as calc_index (:189-191), cells on the room's boundary planes excluded (:118-123).  Ties in the priority queue are broken by
 * insertion order (Julia's Collections.PriorityQueue leaves them unspecified).  path receives the way-points start .. goal in the caller's units (x reso, :259-261);
 * returns their number (the reference's length(rx) = N_as + 1), 0 = no path, -1 = bad arguments / cap too small.  *cost (may be NULL): path cost in grid units. */
int obca_plan_reference_astar3d(const double start[3], const double goal[3], int nob, const double *ox, const double *oy, const double *oz,
                                const double room_min[3], const double room_max[3], double reso, double *path, int cap, int *expansions /* may be NULL */,
                                double *cost /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
