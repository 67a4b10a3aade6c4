/*
 * obca_hip.h -- C ABI of libobca_hip.so: batched OBCA parking signed-distance NLP on AMD MI355X (gfx950).
 *
 * The reference (XiaojingGeorgeZhang/OBCA) has no FFI of its own; its boundary for this path is three positional Julia
 * functions.  Each entry point below is what a `ccall` from a Julia shim binds to replace one of them:
 *
 *   obca_parking_signed_dist_batch  <->  ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)
 *                                        AutonomousParking/ParkingSignedDist.jl:29-314  (call site main.jl:269)
 *   obca_dualmult_ws_batch          <->  DualMultWS(N,nOb,vOb,A,b,rx,ry,ryaw)   AutonomousParking/DualMultWS.jl:29-86
 *                                        (call site ParkingSignedDist.jl:219; `ego` is an explicit argument here, the
 *                                        reference reads a global, DualMultWS.jl:39-45)
 *
 * Conventions: all arrays are caller-allocated host memory, fp64, "stage-contiguous" = exactly the memory of the
 * reference's column-major Julia arrays (x is 4 x (N+1): X1,Y1,psi1,v1,X2,...), with the batch as the slowest dimension,
 * so B=1 is the reference layout.  Per-instance obstacle sets are packed back to back: instance i has nOb[i] obstacles with
 * row counts vOb[ob_off[i] .. ob_off[i]+nOb[i]) and its M_i = sum of those rows of A (2 doubles per row: A[k,1],A[k,2]) and b
 * start at row row_off[i]; ob_off / row_off are the running sums (the library computes them).  lWS/lp are M_i x (N+1),
 * nWS/np are 4 nOb_i x (N+1), packed per instance in the same order.
 * Return value 0 = the call executed (per-instance outcome is in exitflag[] / info[]); negative = API or device error,
 * text in obca_last_error().  No exceptions cross the boundary.  One context is used by one host thread at a time.
 * The library never falls back to a CPU path: without a usable gfx950 device obca_create fails.
 */
#ifndef OBCA_HIP_H
#define OBCA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* Limits of the parking entry points.  The reference has none (obstHrep.jl:31-102 emits one row per polygon edge, main.jl:251 takes whatever horizon the planner gives); here an
 * instance is solved by ONE wavefront whose problem header and sweep buffers live in the CU's LDS, and the (stage, obstacle) block code is compiled for fixed row counts:
 *   OBCA_NMAX   -- horizon: 128 = two stages for each of the 64 lanes of the forward sweep (one lane per stage pair; a longer horizon needs a second round of pair maps,
 *                  not built).  The sweeps' LDS is sized per launch: 5.7 KB + (27 N + 54) x 8 bytes (23 KB at N = 80, 33 KB at N = 128: four instances per CU throughout);
 *                  HBM per instance grows linearly (0.2 MB at N = 80).  The reference's planners give N = 50-110.
 *   OBCA_NOBMAX -- obstacles per instance: 16 bytes of LDS each; the block work is (N + 1) x nOb items on 64 lanes.
 *   OBCA_MMAX   -- half-space rows per instance, all obstacles together: 24 bytes of LDS each (2.0 KB of header at the limits, 0.4 KB used by a 3-obstacle / 5-row instance).
 *   OBCA_VMAX   -- rows of ONE obstacle (polygon edges): the block code exists for <= 2, <= 4 and <= 8 rows (chosen per instance); a 16-row instantiation would keep a
 *                  15 x 15 reduced Hessian per lane in registers and spill most of it -- split such a polygon into convex pieces of <= 8 edges instead.
 * Round 4 lifted the obstacle and row limits from 10 / 40 to these values (the header layout and every buffer follow from the constants). */
#define OBCA_VMAX 8      /* max half-space rows per obstacle */
#define OBCA_MMAX 64     /* max half-space rows per instance (all obstacles together) */
#define OBCA_NOBMAX 16   /* max obstacles per instance */
#define OBCA_NMAX 128    /* max horizon (the reference's planners give N ~ 50-110) */

typedef struct obca_ctx obca_ctx;
typedef struct obca_batch obca_batch;

/* Interior-point options.  The solver behind them is IPOPT's Algorithm A (monotone barrier, filter line search, inertia-correction ladder, alpha_for_y = min) on a
 * structured KKT solve, with the option values of the reference's IPOPT call (ParkingSignedDist.jl:41-43) + IPOPT's defaults.
 * Three IPOPT semantics the reference runs with are switches of the parking kernels:
 *   max_soc  -- the second-order correction (A-5.5..A-5.9 of Waechter & Biegler: up to max_soc corrections with kappa_soc = 0.99 after a rejected first trial step that
 *               did not reduce the constraint violation; IPOPT's default is 4);
 *   recalc_y -- recalc_y = "yes" (ParkingSignedDist.jl:41, ParkingDist.jl:41): the equality multipliers are replaced by their least-squares estimate (the structured solve
 *               with H := I) whenever the accepted iterate's constraint violation is below recalc_y_feas_tol = 1e-6;
 *   lsq_init -- IPOPT's default initial multipliers: the same least-squares estimate at the starting point (constr_mult_init_max = 1e3) instead of y0 = 0.
 * With any of them on, the kernels follow the CPU checker's option of the same name iteration for iteration -- on the FULL bench batches of BASELINE configs 2 / 3 / 5
 * (1 024 + 2 048 + 4 096 instances, all three switches on both sides: every exit flag equal, 0-2 iteration counts differ per batch, tests/test_gpu_parity.py,
 * profiles/r06_parity_census_reference_options_config{2,3,5}.txt; which switch moves a solve into another local solution and what each costs: profiles/r06_options_census.txt).
 *
 * Two option sets, and which one is the default where (measured on one MI355X: round 4 profiles/r04_steps/r04_bench_step1.json, r04_bench_config{3,5}.json; round 5
 * profiles/r05_bench.json, r05_bench_config{3,5}.json):
 *   obca_reference_opts -- the reference's IPOPT configuration as far as the kernels carry it: max_soc = 4, recalc_y = 1, lsq_init = 1, restoration = 1.  The default of the DROP-IN functions
 *                          that carry the reference's names (Julia: OBCAHip.ParkingSignedDist / ParkingDist; Python: obca_amd.ParkingSignedDist / ParkingDist).
 *   obca_default_opts   -- the switches off: the library's THROUGHPUT defaults, what a NULL `opts` means in every entry point below.  (bench.py's `value` runs
 *                          obca_reference_opts since round 5; the throughput set is its secondary leg config.fast_options.)
 * The numbers behind that split: both settings solve every instance of the three bench batches (identical exit flags, every solution passes the a-posteriori checker);
 * the IPOPT configuration costs 6 / 5 / 15 % more iterations and 12 x the inertia-correction rungs (the least-squares start leaves an indefinite Lagrangian Hessian early on):
 * 254.5 k -> 197.7 k, 152.9 k -> 123.0 k, 125.6 k -> 97.2 k solves/s on configs 2 / 3 / 5 (profiles/r04_bench*.json; round 6, sixteen batches in flight: 287.9 k -> 232.3 k, 165.1 k -> 136.8 k, 131.4 k -> 99.8 k, profiles/r06_bench*.json); and 13 of 1 024, 288 of 2 048, 58 of 4 096 instances end in ANOTHER local solution
 * of the non-convex NLP than with the switches off (states / inputs beyond 1e-3, or the objective beyond 1e-4 relative: bench.py, config.ipopt_options).  Neither set of local
 * solutions can be checked against IPOPT itself here (no Julia / IPOPT in the image): the drop-ins run the configuration that is the reference's by construction, the
 * throughput entry points the one that is a fifth cheaper; every bench line reports both.  (Quadcopter, pipelined: 36.7 k -> 31.4 k solves/s with its three switches.)
 * NOT in the kernels: IPOPT's GENERAL restoration phase (both kernels carry a block feasibility restoration in its place: `restoration` below; the quadcopter kernel's is always
 * on), kappa_d damping (the dense un-reformulated pin oracle/ipm_ref80.py carries it and lands on the same solutions to 1e-6: it does not move them), the watchdog.  IPOPT's gradient-based scaling scales no row
 * of these NLPs (half-space rows enter with unit length besides) and the objective by 1 (parking) / 100 / 2 100 (quadcopter: opts.obj_scaling).  The quadcopter kernel carries
 * max_soc, lsq_init and obj_scaling (obca_quadcopter_reference_opts; its reference call sets recalc_y = "no",
 * and its entry points refuse an option record that sets recalc_y).  DESIGN.md sections 2, 9. */
typedef struct obca_opts {
    double tol; int max_iter;
    double mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac;
    double dw_min, dw0, dw_max, kw_inc0, kw_inc, kw_dec, dc_bar, kappa_c;
    double gamma_theta, gamma_phi, delta, s_theta, s_phi, eta_phi, gamma_alpha, s_max, kappa_sigma;
    double constr_viol_tol, dual_inf_tol, compl_inf_tol, rho_term;
    int max_soc;      /* second-order correction trials per iteration (IPOPT max_soc; its default is 4): 0 = off (obca_default_opts), 4 in obca_reference_opts / obca_quadcopter_reference_opts */
    int recalc_y;     /* 1: recalc_y = "yes" as the reference sets it (ParkingSignedDist.jl:41; recalc_y_feas_tol 1e-6): least-squares equality multipliers whenever the
                         iterate's constraint violation is below 1e-6; 0 = off, the default of obca_default_opts; parking kernels only */
    int lsq_init;     /* 1: IPOPT's initial equality multipliers -- the least-squares estimate at the starting point, kept if its max-norm is <= constr_mult_init_max = 1e3;
                         0 = y0 = 0, the default of obca_default_opts (the reference runs IPOPT's default, i.e. 1); parking kernels only */
    int obj_scaling;  /* 1: IPOPT's gradient-based scaling of the objective (nlp_scaling_method default, nlp_scaling_max_gradient = 100): the algorithm runs on sf * f with
                       * sf = 100 / max(100, |grad f(start)|_inf); dual_inf_tol / compl_inf_tol are tested on the unscaled quantities, the reported objective is unscaled.
                       * Quadcopter kernel: sf = 100 / 2 100 at the reference's start (the slack penalty 1e2 + 2e3 * 1), set by obca_quadcopter_reference_opts.  Parking kernels: the
                       * gradient at the reference's own start is the slack penalty 1e2 exactly (sf = 1), but a caller's start (small Ts, jumpy uWS, supplied slacks) can exceed it
                       * and the parking kernels do not scale: the parking entry points REFUSE obj_scaling = 1 (rc -1) instead of ignoring it.  0 = off (the defaults) */
    int restoration;  /* Block feasibility restoration, the stand-in for IPOPT's restoration phase (which the reference leans on: ParkingSignedDist.jl:228-231).  DualMultWS returns
                       * lambda = mu = 0 for a pose that touches or penetrates an obstacle (distance 0), and the signed-distance NLP started there has |A'lam|^2 = 0 against the equality
                       * |A'lam|^2 == 1 with a vanishing row gradient: a rank-deficient start the interior point does not leave.  A (stage, obstacle) block with |A'lam|^2 < 1/4 gets
                       * the feasible dual of the obstacle's best edge (lam = e_s, mu = the non-negative split of -R'a_s; the separation row then holds the signed distance along that
                       * edge normal, its penalised slack absorbs a penetration).  1: at the start of an attempt AND where IPOPT would enter restoration (failed line search, inertia
                       * ladder exhausted; then also: barrier restart, empty filter; at most 3 per attempt); 2: the latter only (test knob); 0 = off (obca_default_opts).
                       * obca_reference_opts sets 1.  An instance whose warm start stays clear of every obstacle at every stage has no degenerate block and is solved to the same bits with
                       * or without.  Planned warm starts do graze obstacles (distance 0 at a stage): 29 of the 1 024 config-2 bench instances, 23 of 1 024 (config 5), 77 of 512
                       * (config 3: the Hybrid A* path touches a corner) carry such a block.  Measured on the oracle: config 2 / 5: the same solutions, the longest solve of the batch
                       * 74 -> 64 iterations; config 3: 60 of 512 instances end in ANOTHER local solution (58 of them lower in the objective) after 37 instead of 55 iterations;
                       * 64 corridor instances whose wedges intrude 0.05 / 0.15 / 0.3 m into the warm start's body: 57 / 45 / 29 solved without, 64 / 64 / 63 with.
                       * The quadcopter kernel always carried its own block restoration (the reference's start lambda = 0.05 is rank-deficient); it ignores this field. */
} obca_opts;

int obca_create(obca_ctx **out, int device);
/* A context over several GPUs of the node ("the context owns devices and streams"): devices = NULL or ndev <= 0 takes every
 * visible device.  The host-pointer entry points below cut their batch into chunks and hand them to the devices through a work queue
 * (solve times are heavy-tailed: a static slice per device would wait for the unluckiest one); every device runs OBCA_SLOTS (default 8, at most 16)
 * chunks at a time on streams of their own, so that the PCIe transfers and the host-side packing of one chunk overlap the solves of the
 * others.  Instances are independent: no collective touches the data path.  Results do not depend on the device count, the chunk size
 * (OBCA_CHUNK, default = the 1 024 instances resident on one GPU: four per CU) or the slot count: every instance is solved by one workgroup either way.
 * The device-resident obca_batch_* / obca_quad_batch_* calls of a multi-device context run on its first device. */
int obca_create_multi(obca_ctx **out, const int *devices, int ndev);
int obca_device_count(const obca_ctx *ctx);          /* devices this context drives */
int obca_visible_device_count(void);                 /* HIP devices visible to the process */
int obca_destroy(obca_ctx *ctx);
const char *obca_last_error(const obca_ctx *ctx);   /* ctx may be NULL: error of the last failed obca_create */
int obca_default_opts(obca_opts *o);      /* throughput defaults (the three IPOPT switches off): what opts == NULL means */
int obca_reference_opts(obca_opts *o);    /* the reference's IPOPT configuration: max_soc = 4, recalc_y = 1, lsq_init = 1, restoration = 1 (see above) */
int obca_device_name(const obca_ctx *ctx, char *buf, int buflen);

/* ---- synchronous host-pointer API (what the Julia shim calls) ---- */
int obca_dualmult_ws_batch(obca_ctx *ctx, int B, int N, const double ego[4], const int *nOb, const int *vOb, const double *A,
                           const double *b, const double *rx, const double *ry, const double *ryaw /* (N+1) x B each */,
                           double *lWS, double *nWS, double *d /* nOb_i x (N+1) packed, may be NULL */);

int obca_parking_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts /* B */, double L, const double ego[4],
                                   const double XYbounds[4], int fixTime, const double *x0 /* 4 x B */, const double *xF /* 4 x B */,
                                   const int *nOb /* B */, const int *vOb, const double *A, const double *b, const double *rx,
                                   const double *ry, const double *ryaw, const double *xWS /* 4 x (N+1) x B */,
                                   const double *uWS /* 2 x N x B */, const double *lWS, const double *nWS /* NULL: run DualMultWS */,
                                   const obca_opts *opts /* NULL: defaults */, double *xp, double *up, double *timeScale /* (N+1) x B */,
                                   int *exitflag /* B */, double *lp, double *np, double *slp /* nOb_i x (N+1) packed, may be NULL */,
                                   double *info /* 8 x B {status,iters,objective,pinf,dinf,mu,nreg,exitflag}, may be NULL */);

/* ParkingDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS) -- AutonomousParking/ParkingDist.jl:29-315 (call site
 * main.jl:258): the collision-free sibling -- no penetration slack, |A'lam|^2 <= 1, weight 0.5 on a^2 (:87), exit flag per :245-289
 * (incl. the inverted feasibility check after a failed retry).  Same argument conventions as the signed-distance call. */
int obca_parking_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double L, const double ego[4], const double XYbounds[4],
                            int fixTime, const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A,
                            const double *b, const double *rx, const double *ry, const double *ryaw, const double *xWS,
                            const double *uWS, const double *lWS, const double *nWS, const obca_opts *opts, double *xp, double *up,
                            double *timeScale, int *exitflag, double *lp, double *np, double *info);

/* ---- device-resident batch API (benchmarks, receding-horizon callers): upload once, solve many times ---- */
int obca_batch_create(obca_ctx *ctx, int B, int N, obca_batch **out);
int obca_batch_destroy(obca_batch *bt);
int obca_batch_set_formulation(obca_batch *bt, int dist);   /* 0 = ParkingSignedDist (default), 1 = ParkingDist; call before obca_batch_upload */
int obca_batch_upload(obca_batch *bt, const double *Ts, double L, const double ego[4], const double XYbounds[4], int fixTime,
                      const double *x0, const double *xF, const int *nOb, const int *vOb, const double *A, const double *b,
                      const double *rx, const double *ry, const double *ryaw, const double *xWS, const double *uWS,
                      const double *lWS, const double *nWS);
int obca_batch_solve(obca_batch *bt, const obca_opts *opts);   /* asynchronous on the context's stream: warm start -> (DualMultWS) -> IPM */
int obca_batch_sync(obca_batch *bt);
/* receding-horizon restart (not in the reference): replace the uploaded warm start by the last solution advanced by `shift`
 * stages (x, u, lambda, mu and the tracking reference move up, the tail repeats the goal stage; t = 1, sl = 0), entirely on the device; the new
 * initial state is x0_new (4 x B host, measured state) or, if NULL, stage `shift` of the solution.  The next obca_batch_solve starts from it. */
int obca_batch_shift_warm_start(obca_batch *bt, int shift, const double *x0_new);
int obca_batch_kernel_ms(obca_batch *bt, float *ipm_ms, float *dualws_ms);   /* HIP-event durations of the last solve */
/* how the last obca_batch_solve ran the interior-point kernel: 1 launch, or the two-launch schedule (a slice of `slice_passes` factorisation
 * passes for every instance, then the parked solves hardest-first; used when the batch exceeds the resident capacity of the GPU, results are
 * bit-identical either way; environment OBCA_SLICE_PASSES=0 disables it).  ipm_ms of obca_batch_kernel_ms covers all launches. */
int obca_batch_last_schedule(const obca_batch *bt, int *ipm_launches, int *slice_passes);
int obca_batch_download(obca_batch *bt, double *xp, double *up, double *timeScale, int *exitflag, double *lp, double *np,
                        double *slp, double *info);
int obca_batch_scratch_bytes(const obca_batch *bt, long long *bytes);
#ifdef OBCA_PROFILE      /* profiling build only (libobca_hip_prof.so: per-phase shader clocks, tools/phase_profile.py); not an entry point of the product library */
int obca_batch_debug_phase_cycles(obca_batch *bt, double *out /* B x 16 */);
#endif

/* ---- quadcopter path:  QuadcopterSignedDist(x0,xF,N,Ts,R,ob1,ob2,ob3,ob4,ob5,xWS,uWS,timeWS)
 *      QuadcopterNavigation/QuadcopterSignedDist.jl:25-298 (call site mainQuadcopter.jl:152).
 * x is 12 x (N+1) stage-contiguous, u 4 x N; ob is 6 x 5 per instance: ob1..ob5 back to back, each [xmax,ymax,zmax,-xmin,-ymin,-zmin]
 * (the `b` of A = [I;-I], :162-166); lp is 30 x (N+1): [l1;l2;l3;l4;l5] stacked as the reference returns it (:295).
 * uWS is accepted for signature parity and ignored like the reference does (:202 starts every input at the hover speed).
 * dual_ws != 0 starts the multipliers at the closed-form point-to-box dual solution (recommended).  dual_ws = 0 is the reference's own start (lambda = 0.05,
 * :204-208): its Jacobian is rank deficient there, IPOPT leaves the point through its restoration phase, this solver through a block feasibility restoration that
 * stands in for it (closed-form distance duals at the current positions, at most three times per solve).
 * exitflag: 1 = solved, 2 = solved but sum(slack) > 1e-3 (:285-288), 0 = failed.  max_iter default 3000, see obca_quadcopter_default_opts. */
#define OBCA_QUAD_NMAX 128   /* mainQuadcopter.jl:116-131: the A* path on the 1.0 grid from x = 10 to 90 gives N_as >= 80 */
typedef struct obca_quad_batch obca_quad_batch;
int obca_quadcopter_default_opts(obca_opts *o);
/* the reference's IPOPT configuration for this call as far as the quadcopter kernel carries it: the defaults above + max_soc = 4 (IPOPT's default second-order
 * correction, A-5.5 .. A-5.9) + lsq_init = 1 (IPOPT's default least-squares initial multipliers, kept if <= 1e3) + obj_scaling = 1 (IPOPT's default gradient-based scaling:
 * the objective factor 100 / 2 100 on this NLP); recalc_y stays off -- QuadcopterSignedDist.jl:29 sets recalc_y = "no", and opts.recalc_y != 0 is refused by the quadcopter
 * entry points rather than ignored. */
int obca_quadcopter_reference_opts(obca_opts *o);
int obca_quadcopter_signed_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts /* B */, double R, const double *x0 /* 12 x B */,
                                      const double *xF /* 12 x B */, const double *ob /* 6 x 5 x B */, const double *xWS /* 12 x (N+1) x B */,
                                      const double *uWS /* ignored, may be NULL */, const double *timeWS /* B */, int dual_ws,
                                      const obca_opts *opts /* NULL: defaults */, double *xp, double *up, double *timeScale /* (N+1) x B */,
                                      int *exitflag /* B */, double *lp /* 30 x (N+1) x B */, double *slack /* 5 x (N+1) x B, may be NULL */,
                                      double *info /* 8 x B, may be NULL */);
/* QuadcopterDist(x0,xF,N,Ts,R,ob1..ob5,xWS,uWS,timeWS) -- QuadcopterNavigation/QuadcopterDist.jl:25-282 (call site mainQuadcopter.jl:145):
 * the collision-free sibling: no slack variable (:47,:65,:165...), x[10] in [-1.5, 3] (:88), exit flag 0/1 only. */
int obca_quadcopter_dist_batch(obca_ctx *ctx, int B, int N, const double *Ts, double R, const double *x0, const double *xF,
                               const double *ob, const double *xWS, const double *uWS /* ignored */, const double *timeWS, int dual_ws,
                               const obca_opts *opts, double *xp, double *up, double *timeScale, int *exitflag, double *lp, double *info);
int obca_quad_batch_create(obca_ctx *ctx, int B, int N, obca_quad_batch **out);
int obca_quad_batch_destroy(obca_quad_batch *bt);
int obca_quad_batch_upload(obca_quad_batch *bt, const double *Ts, double R, const double *x0, const double *xF, const double *ob,
                           const double *xWS, const double *timeWS, int dual_ws, int dist /* 1: QuadcopterDist formulation */);
int obca_quad_batch_solve(obca_quad_batch *bt, const obca_opts *opts);   /* asynchronous on the context's stream */
int obca_quad_batch_sync(obca_quad_batch *bt);
int obca_quad_batch_kernel_ms(obca_quad_batch *bt, float *ipm_ms);
int obca_quad_batch_download(obca_quad_batch *bt, double *xp, double *up, double *timeScale, int *exitflag, double *lp, double *slack,
                             double *info);
int obca_quad_batch_scratch_bytes(const obca_quad_batch *bt, long long *bytes);
#ifdef OBCA_PROFILE
int obca_quad_batch_debug_phase_cycles(obca_quad_batch *bt, double *out /* B x 16 */);
#endif

#ifdef __cplusplus
}
#endif
#endif
