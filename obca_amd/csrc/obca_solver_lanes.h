// obca_solver_lanes.h -- part of obca_solver.h (included from there, inside namespace obca; not a stand-alone header):
// reductions over the lanes of an instance, the ordered sum of the condensed obstacle contributions, small per-lane helpers, accepting a step.

// ---------------------------------------------------------------- reductions over the lanes of the instance
// (Two-wavefront instances -- the quadcopter kernel, NT = 128 -- first fold the second wavefront's
// slots onto the first.)  A 64-lane butterfly in ASCENDING distance (1, 2, 4, 8, 16, 32).  The first
// four exchanges stay inside a row of 16 lanes and run as DPP moves on the vector ALU (quad permutes, then half-row and row mirrors:
// once every lane of a quad / half-row holds the same partial result, the mirrored partner carries exactly what the xor partner
// would); only distances 16 and 32 cross rows and go through ds_bpermute.  An LDS exchange costs a ~100-clock round trip that the
// compiler serialises per reduction, and a pass holds some thirty reductions.  The emulation pairs lanes i and i^o in the same
// order, so its results are bit-identical (sum and max are commutative).
#ifdef OBCA_EMU
#define RED_IMPL(NAME, COMB)                                                                                     \
    template <int NT> OBCA_FN double NAME(const double *r) {                                                     \
        double a[64], b[64];                                                                                     \
        for (int i = 0; i < 64; i++) { a[i] = r[i]; if (NT > 64) { double w = r[i + 64 * (NT > 64)], v = a[i]; a[i] = COMB; } } \
        for (int o = 1; o < 64; o <<= 1) {                                                                       \
            for (int i = 0; i < 64; i++) { double v = a[i], w = a[i ^ o]; b[i] = COMB; }                         \
            for (int i = 0; i < 64; i++) a[i] = b[i];                                                            \
        }                                                                                                        \
        return a[0];                                                                                             \
    }
#else
OBCA_FN double readlane_f64(double v, const int l) {   // value of lane l (a constant) as a wave-uniform scalar
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
OBCA_FN double dpp_f64(double v) {   // every lane active (the reductions are called from uniform control flow)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#define RED_IMPL(NAME, COMB)                                                                                     \
    template <int NT> OBCA_FN double NAME(const double *r) {                                                     \
        double v = r[threadIdx.x & 63], w;                                                                       \
        if (NT > 64) { w = r[(threadIdx.x & 63) + 64 * (NT > 64)]; v = COMB; }                                   \
        w = dpp_f64<0xB1>(v); v = COMB;          /* quad_perm [1,0,3,2]  : i ^ 1 */                               \
        w = dpp_f64<0x4E>(v); v = COMB;          /* quad_perm [2,3,0,1]  : i ^ 2 */                               \
        w = dpp_f64<0x141>(v); v = COMB;         /* row_half_mirror      : stands in for i ^ 4 */                 \
        w = dpp_f64<0x140>(v); v = COMB;         /* row_mirror           : stands in for i ^ 8 */                 \
        w = __shfl_xor(v, 16, 64); v = COMB;                                                                     \
        w = __shfl_xor(v, 32, 64); v = COMB;                                                                     \
        return v;                                                                                                \
    }
#endif
// the same butterflies on per-lane REGISTER values (one-wavefront instances: nothing goes through LDS).  In the host emulation a per-lane value that lives
// across the lanes' loop is an array over the lanes (OBCA_NL = 64), on the GPU it is one register (OBCA_NL = 1).
#ifdef OBCA_EMU
#define WRED_IMPL(NAME, COMB)                                                                                    \
    OBCA_FN double NAME(const double (&r)[OBCA_NL]) {                                                            \
        double a[64], b[64];                                                                                     \
        for (int i = 0; i < 64; i++) a[i] = r[i];                                                                \
        for (int o = 1; o < 64; o <<= 1) {                                                                       \
            for (int i = 0; i < 64; i++) { double v = a[i], w = a[i ^ o]; b[i] = COMB; }                         \
            for (int i = 0; i < 64; i++) a[i] = b[i];                                                            \
        }                                                                                                        \
        return a[0];                                                                                             \
    }
#else
#define WRED_IMPL(NAME, COMB)                                                                                    \
    OBCA_FN double NAME(const double (&r)[OBCA_NL]) {                                                            \
        double v = r[0], w;                                                                                      \
        w = dpp_f64<0xB1>(v); v = COMB;                                                                          \
        w = dpp_f64<0x4E>(v); v = COMB;                                                                          \
        w = dpp_f64<0x141>(v); v = COMB;                                                                         \
        w = dpp_f64<0x140>(v); v = COMB;                                                                         \
        w = __shfl_xor(v, 16, 64); v = COMB;                                                                     \
        w = __shfl_xor(v, 32, 64); v = COMB;                                                                     \
        return v;                                                                                                \
    }
#endif
// sum over each quad of lanes (4 q .. 4 q + 3), left in all four of them: two DPP exchanges
#ifdef OBCA_EMU
OBCA_FN void wquad_sum(const double (&r)[OBCA_NL], double (&out)[OBCA_NL]) {
    double a[64];
    for (int i = 0; i < 64; i++) a[i] = r[i] + r[i ^ 1];
    for (int i = 0; i < 64; i++) out[i] = a[i] + a[i ^ 2];
}
#else
OBCA_FN void wquad_sum(const double (&r)[OBCA_NL], double (&out)[OBCA_NL]) { double v = r[0]; v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); out[0] = v; }
#endif
// value of lane l (a constant) of a per-lane variable as a wave-uniform scalar (v_readlane; the emulation keeps per-lane variables as arrays over the lanes)
#ifdef OBCA_EMU
#define WV_READLANE(x, l) ((x)[l])
#else
#define WV_READLANE(x, l) readlane_f64((x)[0], (l))
#endif
WRED_IMPL(wred_sum, (v + w))
WRED_IMPL(wred_max, ((w > v || w != w) ? w : v))      // NaN-propagating max
WRED_IMPL(wred_min, ((w < v) ? w : v))
RED_IMPL(red_sum_t, (v + w))
RED_IMPL(red_max_t, ((w > v || w != w) ? w : v))      // NaN-propagating max
RED_IMPL(red_min_t, ((w < v) ? w : v))
OBCA_FN double red_sum(const double *r) { return red_sum_t<OB_NT>(r); }
OBCA_FN double red_max(const double *r) { return red_max_t<OB_NT>(r); }
OBCA_FN double red_min(const double *r) { return red_min_t<OB_NT>(r); }

// Sum of the condensed contributions (12 doubles) of the obstacles of a stage, in the
// order of the obstacles -- the sums are the same bits in every run, on every box.
// Items are stage-major (item = k nOb + j), so the lanes of one round that belong to
// a stage are neighbours: position p = min(j, lane) within the stage's run of lanes.
// A running sum walks down the run, one lane per step (wave_shr:1 moves it to the next
// lane; nOb - 1 uniform steps, all lanes take part in the moves, only the lane whose
// turn it is adds); the last lane of the run stores the 12 sums.  A stage whose obstacles
// straddle two rounds is continued: lane 0 of the next round starts from the stored
// partial sums (LDS traffic of one wavefront is in order, the rounds in program order).
// The first obstacle of a stage starts from +0, as the emulation's cleared cell does.
// (Round 4 used ds_add_f64 here: up to 16 lanes of one instruction on one address, relying
// on the hardware serving them in lane order -- nothing documents that, fp64 addition
// is not associative, and the driver's round-4 GPU run saw two runs of the same inputs
// differ.  tools/micro/lds_atomic_order.hip probes the order)
#ifdef OBCA_EMU
OBCA_FN void obs_sum_ordered(double *ocs, const ObsCond &cd, int k, int j, bool on, int nOb, int lane) {
    (void)j; (void)nOb; (void)lane;
    if (!on) return;
    double *o = ocs + (size_t)k * OB_OC;
    for (int i = 0; i < 6; i++) o[i] += cd.Hpp[i];
    for (int i = 0; i < 3; i++) { o[6 + i] += cd.gz[i]; o[9 + i] += cd.gcorr[i]; }
}
#else
OBCA_FN void obs_sum_ordered(double *ocs, const ObsCond &cd, int k, int j, bool on, int nOb, int lane) {
    double c[OB_OC], run[OB_OC];
#pragma unroll
    for (int i = 0; i < 6; i++) c[i] = cd.Hpp[i];
#pragma unroll
    for (int i = 0; i < 3; i++) { c[6 + i] = cd.gz[i]; c[9 + i] = cd.gcorr[i]; }
    double *o = ocs + (size_t)k * OB_OC;
    const int p = j < lane ? j : lane;
    LDS_SYNC();                                        // the partial sums the previous round stored are visible (and the compiler keeps the order)
    const bool cont = on && lane == 0 && j > 0;        // the stage began in the previous round
#pragma unroll
    for (int i = 0; i < OB_OC; i++) run[i] = 0.0;
    if (cont) {
#pragma unroll
        for (int i = 0; i < OB_OC; i++) run[i] = o[i];
    }
#pragma unroll
    for (int i = 0; i < OB_OC; i++) run[i] += c[i];
    for (int s = 1; s < nOb; s++) {                    // uniform
        double t[OB_OC];
#pragma unroll
        for (int i = 0; i < OB_OC; i++) t[i] = dpp_f64<0x138>(run[i]);      // wave_shr:1 -- lane l receives lane l - 1's running sum
        if (on && p == s) {
#pragma unroll
            for (int i = 0; i < OB_OC; i++) run[i] = t[i] + c[i];
        }
    }
    if (on && (j == nOb - 1 || lane == OB_NT - 1)) {   // end of the stage's run in this round (the last item of all is a last obstacle)
#pragma unroll
        for (int i = 0; i < OB_OC; i++) o[i] = run[i];
    }
}
#endif
// the per-instance constants the (stage, obstacle) block code reads, copied into scalar
// registers (as LDS reads they would sit in vector registers for the whole item loop)
OBCA_FN void obs_consts(const Consts &s_, Consts &c) {
    c.N = UNIFORM(s_.N); c.nOb = UNIFORM(s_.nOb); c.M = UNIFORM(s_.M); c.dist = UNIFORM(s_.dist); c.fixTime = UNIFORM(s_.fixTime);
    c.off = UNIFORM_D(s_.off);
#pragma unroll
    for (int i = 0; i < 4; i++) c.g[i] = UNIFORM_D(s_.g[i]);
}
template <int VM>
OBCA_FN void load_obs(const Inst &I, const Shared &sh, const gdbl *z, int k, int j, ObsIn<VM> &in) {
    const Lay &l = sh.l; const int nOb = sh.c.nOb, M = sh.c.M;
    const int r0 = sh.roff[j], v = sh.vOb[j], bo = k * nOb + j;
    in.v = v;
#pragma unroll
    for (int i = 0; i < VM; i++) {
        bool on = i < v;
        in.a1[i] = on ? sh.hdr[PH_A + 2 * (r0 + i)] : 0.0; in.a2[i] = on ? sh.hdr[PH_A + 2 * (r0 + i) + 1] : 0.0;
        in.b[i] = on ? sh.hdr[PH_B + r0 + i] : 0.0;
        in.lam[i] = on ? z[l.lam + k * M + r0 + i] : 1.0; in.zl[i] = on ? z[l.zlam + k * M + r0 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { in.mu[i] = z[l.mu + 4 * bo + i]; in.zm[i] = z[l.zmu + 4 * bo + i]; in.y[i] = z[l.yo + 4 * bo + i]; }
    in.sl = z[l.sl + bo]; in.so = z[l.so + bo]; in.zso = z[l.zso + bo]; in.zs1 = z[l.zs1 + bo];
    in.X = z[l.x + 4 * k]; in.Y = z[l.x + 4 * k + 1]; in.psi = z[l.x + 4 * k + 2];
}

struct B2 { double Sig, gz, gb; };
// (the largest |s z| is not tracked: it is max(|smallest product|, |largest product|), formed once from cmn / cmx where the assembly ends)
OBCA_FN B2 bound2(double v, double lo, double hi, double zL, double zU, double mu, double mult, double &cmn, double &cmx, double &sumz) {
    const double dL = v - lo, dU = hi - v, iL = rcp_nr(dL), iU = rcp_nr(dU);
    B2 r; r.Sig = mult * (zL * iL + zU * iU); r.gz = mult * (-zL + zU); r.gb = mult * mu * (iU - iL);
    double c1 = dL * zL, c2 = dU * zU;
    cmn = fmin(cmn, fmin(c1, c2)); cmx = fmax(cmx, fmax(c1, c2));
    sumz += fabs(zL) + fabs(zU);
    return r;
}
// the same with the running max of |s z| kept by the caller (the quadcopter kernel's stage assembly, obca_quad_solver.h)
OBCA_FN B2 bound2(double v, double lo, double hi, double zL, double zU, double mu, double mult, double &c0, double &cmn, double &cmx, double &sumz) {
    const B2 r = bound2(v, lo, hi, zL, zU, mu, mult, cmn, cmx, sumz);
    c0 = fmax(c0, fmax(fabs((v - lo) * zL), fabs((hi - v) * zU)));
    return r;
}
// Barrier sums.  sum_i log(d_i) is evaluated as log(prod_i d_i) over groups of at most G distances: a double-precision log is a ~2k-clock
// dependent chain for a lone wavefront and there are a dozen per stage / obstacle item, while the product of twelve distances in
// [1e-25, 1e25] stays inside the double range.  A non-positive distance poisons its group (NaN), as its own log would.  The assembly and
// the trial evaluation use the same groups in the same order, so the same point gives the same bits in both.
template <int NN, int G = 12>
OBCA_FN double log_prod(const double (&dd)[NN]) {
    double s_ = 0;
#pragma unroll
    for (int g = 0; g < NN; g += G) {
        double p0 = 1, p1 = 1, mn = 1;
#pragma unroll
        for (int i = g; i < g + G && i < NN; i++) { if (i & 1) p1 *= dd[i]; else p0 *= dd[i]; mn = fmin(mn, dd[i]); }
        const double lg = log(p0 * p1);
        s_ += mn > 0 ? lg : NAN;
    }
    return s_;
}
// the same with running accumulators (two product chains, lower / upper distances) for code that meets its bounds one at a time
struct BarAcc { double p0, p1, mn; };
OBCA_FN void bar_init(BarAcc &a) { a.p0 = a.p1 = a.mn = 1.0; }
OBCA_FN void bar_mul(BarAcc &a, double dlo, double dhi) { a.p0 *= dlo; a.p1 *= dhi; a.mn = fmin(a.mn, fmin(dlo, dhi)); }
OBCA_FN double bar_log(const BarAcc &a) { const double lg = log(a.p0 * a.p1); return a.mn > 0 ? lg : NAN; }
OBCA_FN int hidx(int i, int j) { int a_ = i < j ? i : j, b_ = i < j ? j : i; return a_ * 8 - a_ * (a_ - 1) / 2 + (b_ - a_); }
// Which entries of a stage record exist.  Hessian: pose block (obstacles, tracking),
// the (psi, v, delta, a) block of the bicycle model, the rate terms (w, u) and the
// steering row (w0, delta) -- 19 of 36; everything else is structurally zero and neither stored nor gathered by the backward sweep (-1).
OBCA_FN int as_h(int i, int j) {
    const int a_ = i < j ? i : j, b_ = i < j ? j : i;
    switch (a_ * 8 + b_) {
        case 0: return 0; case 1: return 1; case 2: return 2; case 9: return 3; case 10: return 4; case 18: return 5; case 19: return 6; case 22: return 7;
        case 23: return 8;
        case 27: return 9; case 30: return 10; case 31: return 11; case 36: return 12; case 38: return 13; case 45: return 14; case 47: return 15;
        case 54: return 16;
        case 55: return 17; case 63: return 18; default: return -1;
    }
}
// Jacobian: F_psi does not depend on psi itself beyond the identity, F_v only on a and t
OBCA_FN int as_df(int i, int j) { return i < 2 ? 5 * i + j : (i == 2 ? (j >= 1 ? 9 + j : -1) : (j >= 3 ? 11 + j : -1)); }


// ---------------------------------------------------------------- accepting a step: new bound multipliers
template <int RS = 1>
OBCA_FN double clampz(double zz, double dist, double mu, double ks) { const double q = mu * rcp_nr<RS>(dist), lo = q * rcp_nr<RS>(ks), hi = ks * q; return zz < lo ? lo : (zz > hi ? hi : zz); }
// bound-multiplier step for a lower bound at distance `dist` (upper bound: pass -dv):  z += az (mu/dist - z - z/dist dv)
template <int RS = 1>
OBCA_FN double zstep(double zz, double dist, double dv, double mu, double az) { const double id = rcp_nr<RS>(dist); return zz + az * (mu * id - zz - zz * id * dv); }


