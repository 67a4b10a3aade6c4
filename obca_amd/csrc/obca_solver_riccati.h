// obca_solver_riccati.h -- part of obca_solver.h (included from there, inside namespace obca; not a stand-alone header):
// Riccati backward sweep (three LDS phases per stage) and the wave-level matrix-core helpers the quadcopter sweep uses.

// ---------------------------------------------------------------- Riccati backward sweep
// Stage k is condensed onto (x_k, w_k=u_{k-1}); six right-hand sides (main, t, nu1..4) ride along as extra columns and the
// bilinear constants B(a,b) of the cost-to-go give every entry of the 5x5 (t, nu) border without a forward pass per column.
// Returns 1 if every 2x2 input block is positive definite.
OBCA_FN void pair_of(int p, int &a_, int &b_) {   // p-th pair (a<=b) of the 6 columns, row-major upper triangle
    a_ = (p >= 6) + (p >= 11) + (p >= 15) + (p >= 18) + (p >= 20);
    b_ = p - (6 * a_ - a_ * (a_ - 1) / 2) + a_;
}

// unpacked stage data in LDS (one of two buffers): H (8x8 full), FA' = [Fm | off]' (14x6: FA'[cc * 6 + a]), hc (8x6)
#define SG_H 0
#define SG_FA 64
#define SG_HC 148
#define SG_SIZE 196
// Staged values of a stage: 196, of which 92 are constants of the layout (identity / zero pattern of FA, unused right-hand-side columns of hc):
// those are written ONCE per sweep into both buffers (stage_unpack_constants); the 104 that change with the stage -- H (64, the symmetric entries
// twice), the 24 bicycle-model entries of FA, the 16 gradient / time columns of hc --
// of which 64 can be non-zero (as_h, as_df) -- are gathered per stage, ONE per lane
// (value = kc + rec[idx]); which position a lane serves is tabulated once per solve (Shared::upos).
#define SG_NVAR 64
struct UnpackPlan { int idx, dst; double kc; };
OBCA_FN void stage_unpack_item(int it, int &idx, int &dst, double &fl, double &kc) {     // all 196 positions: what is stored where (fl = 0: the constant kc)
    idx = AS_DD; fl = 0.0; kc = 0.0; dst = SG_SIZE;              // default: harmless gather, store to the pad slot behind the buffer
    if (it < 64) { dst = SG_H + it; if (as_h(it >> 3, it & 7) >= 0) { idx = AS_H + as_h(it >> 3, it & 7); fl = 1.0; } }
    else if (it < 64 + 84) {
        const int e = it - 64, a_ = e / 14, cc = e % 14; dst = SG_FA + cc * 6 + a_;      // FA is staged TRANSPOSED: row cc of FA' = column cc of FA, contiguous
        if (cc < 8) {
            if (a_ < 4) {
                if (cc < 4) kc = (a_ == cc) ? 1.0 : 0.0;
                const int jc = cc == 2 ? 0 : (cc == 3 ? 1 : (cc == 6 ? 2 : (cc == 7 ? 3 : -1)));
                if (jc >= 0 && as_df(a_, jc) >= 0) { idx = AS_DF + as_df(a_, jc); fl = 1.0; }
            } else kc = (cc == a_ + 2) ? 1.0 : 0.0;
        } else if (a_ < 4) { const int col = cc - 8; if (col == 0) { idx = AS_DD + a_; fl = 1.0; } if (col == 1) { idx = AS_DF + as_df(a_, 4); fl = 1.0; } }
    } else if (it < SG_SIZE) {
        const int e = it - 148, i = e / OB_NC, cc = e % OB_NC; dst = SG_HC + e;
        if (cc == 0) { idx = AS_HB + i; fl = 1.0; }
        if (cc == 1 && i >= 2) { idx = AS_HT + i - 2; fl = 1.0; }
    }
}
// The item maps above are irregular (a divergent switch per position), so they are evaluated ONCE per solve into two small LDS tables; a sweep only reads them:
//   upl[lane]    = (idx << 8) | dst | one << 16 : the stage-dependent position this lane gathers (there are exactly SG_NVAR = OB_NT of them)
//   ucn[r][lane] = dst | one << 16, or -1       : the constant positions (+ the pad slot) this lane rewrites at the start of a sweep (the stage buffers share
//                                                  their LDS with the forward sweep's pair maps)
#define SG_NCONST_ROUNDS 3      // (SG_SIZE + 1 - SG_NVAR = 133 constant positions over 64 lanes)
OBCA_FN void init_unpack_table(Shared &sh) {
    PAR(lane) {
        int nv_ = 0, nc_ = 0;
        for (int r = 0; r < SG_NCONST_ROUNDS; r++) sh.ucn[r][lane] = -1;
        for (int it = 0; it <= SG_SIZE; it++) {
            int idx, dst; double fl, kc; stage_unpack_item(it, idx, dst, fl, kc);
            const int one = kc != 0.0 ? (1 << 16) : 0;
            if (fl != 0.0) { if (nv_ == lane) sh.upl[lane] = (idx << 8) | dst | one; nv_++; }
            else { if (nc_ % OB_NT == lane && nc_ / OB_NT < SG_NCONST_ROUNDS) sh.ucn[nc_ / OB_NT][lane] = dst | one; nc_++; }
        }
    }
}
OBCA_FN void stage_unpack_plan(const Shared &sh, int lane, UnpackPlan &p) { const int w = sh.upl[lane]; p.idx = (w >> 8) & 0xff; p.dst = w & 0xff; p.kc = (w >> 16) & 1 ? 1.0 : 0.0; }
OBCA_FN void stage_unpack_constants(const Shared &sh, double *sg, int lane) {     // once per sweep, both buffers (+ the pad slot)
#pragma unroll
    for (int r = 0; r < SG_NCONST_ROUNDS; r++) {
        const int w = sh.ucn[r][lane];
        if (w >= 0) { const double kc = (w >> 16) & 1 ? 1.0 : 0.0; sg[w & 0xffff] = kc; sg[OB_STG + (w & 0xffff)] = kc; }
    }
}
// an independent, branch-free gather; the raw value is only touched at store time
OBCA_FN void stage_unpack_load(const gdbl *rec, const UnpackPlan &p, double &v) { v = rec[p.idx]; }
OBCA_FN void stage_unpack_store(double *sg, const UnpackPlan &p, const double v) { sg[p.dst] = p.kc + v; }

// A dependent fp64 operation costs ~45 clock ticks when an instance runs alone on its CU (one wavefront per SIMD: nothing fills the pipeline;
// tools/micro/lds_barrier_latency.hip), so the short dot products of the sequential sweeps are summed as a tree (depth 4 instead of 7).
OBCA_FN double dot6_tree(double init, double a0, double b0, double a1, double b1, double a2, double b2, double a3, double b3, double a4, double b4,
                         double a5, double b5) {
    const double t0 = fma(a1, b1, a0 * b0), t1 = fma(a3, b3, a2 * b2), t2 = fma(a5, b5, fma(a4, b4, init));
    return (t0 + t1) + t2;
}
OBCA_FN double dot4_tree(double init, const double (&a)[4], const double *b) { return fma(a[1], b[1], a[0] * b[0]) + fma(a[3], b[3], fma(a[2], b[2], init)); }
// NV contiguous, 16-byte aligned doubles from LDS as ds_read_b128
template <int NV>
OBCA_FN void ldv(const double *q, double (&v)[NV]) {
#ifdef OBCA_EMU
    for (int i = 0; i < NV; i++) v[i] = q[i];
#else
    const double2 *q2 = (const double2 *)__builtin_assume_aligned(q, 16);
#pragma unroll
    for (int i = 0; i < NV / 2; i++) { const double2 t = q2[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
#endif
}
#ifndef RIC_D
#define RIC_D 4   // stage records are gathered from HBM this many stages before they are needed (memory latency >> one stage of math)
#endif
// One stage of the sweep on the 64 lanes of the wavefront: three short LDS phases (T
// = P [F|off] + [0|p];  Qhat = [H|hc] + F'T;  eliminate u_k), ONE item per lane and
// phase, wave-local LDS ordering in between (no cross-wavefront synchronisation: the
// instance IS one wavefront).  Every lane runs the SAME straight-line code in every phase:
// what differs between the item kinds of a phase is only where the operands live, and that is a per-lane table of LDS offsets built once per sweep (RicItem).
// A phase costs what its one wavefront ISSUES (a 16-byte LDS read ~16 clocks, an fp64 operation 4, 8 if it depends on the one before: tools/micro/fp64_dependent_latency.hip)
// plus one LDS round trip, so the items are
// cut down to the products that are not structure (rounds 1-3 computed all 96 / 124 / 93 entries, two per lane):
//   * FA = [F | off] has the unit columns 0, 1 (X, Y), the zero columns 4, 5 (the input
//     copy w: x+ does not depend on it) and 10..13 (the nu right-hand sides): the columns
//     0, 1, 4, 5, 10..13 of T are columns of P, zero, or columns of p -- phase B reads them where they are; phase A forms the six others (36 items).
//   * rows 4, 5 and columns 4, 5 of Qhat are [H | hc] itself (F has nothing there): phase
//     C reads them from the stage buffer; rows 0, 1 of Qhat are [H | hc] + T rows 0, 1:
//     for the copied columns that is one more phase-A item each (16), for the others a
//     phase-B item with a unit-vector operand (12).  Phase B: rows psi, v, delta, a over the
//     twelve live columns (48) + those 12 + 4.
//   * P is symmetric: phase C forms the 21 entries i <= c once and stores them twice
//     (exactly symmetric, where rounds 1-3 computed both halves), and the 36 entries of p.
//   * the static parts of the bilinear constants ACCUMULATE in their own slots over the
//     stages (the item's initial value is its previous sum); u1[m][b] = off_m . T(8+b)
//     equals u2[m][b] = off_m . p(b) for b >= 2 (T(8+b) = p(b) there) and is not formed.
//     The dynamic part - Qhat_u(a)' Quu^-1 Qhat_u(b) is not needed before the sweep ends:
//     every stage leaves Qhat_u of its six right-hand sides, Quu and 1 / det in LDS (RIC_BD
//     doubles) and the 21 sums over the stages are formed afterwards, three lanes per pair.
// PIPE = 1: steady state of the software pipeline -- the last phase first retires the gather of stage k-1 (issued RIC_D stages ago into
// nv[..][slot]) into the LDS buffer and re-issues the slot for stage k-1-RIC_D.  Every global load / store is issued unconditionally
// (clamped stage index, dummy slot RS_PAD for the lanes without an item) and the loop has a single exit: with no branch around a
// memory operation the compiler's in-order vmcnt bookkeeping stays exact and old gathers retire without draining the younger ones.
struct RicItem {      // offsets in doubles from the start of Shared
    // phase A: the two operand vectors (4 contiguous doubles each), two initial values, destination; *_sg: bit 0/1/2 = A/B/first initial
    int a_a, a_b, a_i, a_j, a_d, a_sg;
    int b_a, b_b, b_i, b_j, b_d, b_sg;      //          value live in the stage buffer (its parity offset is added at run time); phase B likewise
    // phase C: see riccati_stage (c_sg: bits 0..4 = x6, x7, q6, q7, base live in the stage buffer)
    int c_x6, c_x7, c_q6, c_q7, c_base, c_sg, c_d1, c_d2, c_bd, c_rv, c_rk0, c_rk1, c_dump;
};
#define RIC_BD 16       // per stage: Qhat_u (rows 6, 7) of the six right-hand sides, then q00, q10, q11, 1 / det
// where phase C finds Qhat[r][c]: rows / columns 4, 5 are [H | hc] in the stage buffer
OBCA_FN int ric_qsrc(int oQ, int oSG, int r, int c, int bit, int &sg) {
    if (r == 4 || r == 5 || c == 4 || c == 5) { sg |= bit; return oSG + (c < 8 ? SG_H + r * 8 + c : SG_HC + r * OB_NC + (c - 8)); }
    return oQ + r * 14 + c;
}
OBCA_FN void ric_item(const Shared &sh, int lane, RicItem &p) {
    const double *L = (const double *)&sh; const RicLds &rl = ric_lds(sh);
    const int oPn = (int)(rl.Pn - L), opn = (int)(rl.pn - L), oQ = (int)(rl.Qhat - L), osB = (int)(rl.sB - L), oT = (int)(rl.TT - L),
              oSG = (int)(ric_sg0(sh) - L), oZ = (int)(&rl.zero - L), oZ6 = (int)(rl.zero6 - L), oD = (int)(&rl.dump - L), oD4 = (int)(rl.dump4 - L);
    const int S6[6] = {2, 3, 6, 7, 8, 9}, R8[8] = {0, 1, 4, 5, 10, 11, 12, 13}, I4[4] = {2, 3, 6, 7}, C12[12] = {0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13};
    // A: items 0..35 T[a][cc] = [cc >= 8] p[a][cc-8] + P[a][:] . FA[:][cc] for the six
    // live columns (stored as T'[cc][a]);  36..47 u2[m][b] += FA[:][8+m] . p[:][b];
    //    48..63 Qhat[a][cc] = [H | p][a][cc] + P[a][:] . FA[:][cc] for a = 0, 1 and the copied columns (FA[:][cc] is a unit vector or zero there)
    // (F acts through its rows 0..3 only -- the bicycle model -- plus the selector rows w+
    // = u, whose coefficient is exactly 1: every product is a 4-term dot product with up
    //  to two initial values, and the dependency chain of an item is three operations deep instead of four)
    p.a_a = oZ6; p.a_b = oZ6; p.a_i = oZ; p.a_j = oZ; p.a_d = oD; p.a_sg = 0;
    if (lane < 36) { const int cc = S6[lane / 6], a_ = lane % 6; p.a_a = oPn + a_ * 6; p.a_b = oSG + SG_FA + cc * 6; p.a_sg = 2;
                     p.a_i = cc < 8 ? oZ : opn + (cc - 8) * 6 + a_; p.a_j = cc == 6 ? oPn + a_ * 6 + 4 : (cc == 7 ? oPn + a_ * 6 + 5 : oZ);
                     p.a_d = oT + cc * 6 + a_; }
    else if (lane < 48) { const int m = (lane - 36) / 6, b_ = (lane - 36) % 6; p.a_a = oSG + SG_FA + (8 + m) * 6; p.a_sg = 1; p.a_b = opn + b_ * 6; p.a_i = p.a_d = osB + 12 + m * 6 + b_; }
    else { const int a_ = (lane - 48) / 8, cc = R8[(lane - 48) % 8]; p.a_a = oPn + a_ * 6; p.a_b = oSG + SG_FA + cc * 6; p.a_sg = 2; p.a_d = oQ + a_ * 14 + cc;
           if (cc < 8) { p.a_i = oSG + SG_H + a_ * 8 + cc; p.a_sg |= 4; } else p.a_i = opn + (cc - 8) * 6 + a_; }
    // B: items 0..47 Qhat[i][cc] = [H | hc][i][cc] + FA[:][i] . T[:][cc] for the rows psi,
    // v, delta, a and the twelve live columns;  48..59 the same for rows X, Y and the six
    //    columns phase A formed;  60..63 u1[m][b] += FA[:][8+m] . T[:][8+b], b = 0, 1.
    //    T[:][cc] is read where it lives: a row of P (symmetric), a column of p', or T'
    p.b_a = oZ6; p.b_b = oZ6; p.b_i = oZ; p.b_j = oZ; p.b_d = oD; p.b_sg = 0;
    if (lane < 60) {
        const int i = lane < 48 ? I4[lane / 12] : (lane - 48) / 6, cc = lane < 48 ? C12[lane % 12] : S6[(lane - 48) % 6];
        p.b_a = oSG + SG_FA + i * 6; p.b_sg = 1 | 4;
        p.b_b = cc < 2 ? oPn + cc * 6 : (cc >= 10 ? opn + (cc - 8) * 6 : oT + cc * 6);
        // the selector rows of F: + T[4][cc] for the delta row, + T[5][cc] for the a row
        p.b_j = i == 6 ? p.b_b + 4 : (i == 7 ? p.b_b + 5 : oZ);
        p.b_i = oSG + (cc < 8 ? SG_H + i * 8 + cc : SG_HC + i * OB_NC + (cc - 8)); p.b_d = oQ + i * 14 + cc;
    } else { const int m = (lane - 60) / 2, b_ = (lane - 60) % 2; p.b_a = oSG + SG_FA + (8 + m) * 6; p.b_sg = 1; p.b_b = oT + (8 + b_) * 6;
    p.b_i = p.b_d = osB + m * 6 + b_; }
    // C: value = base + (X6 n0 + X7 n1) / det with (n0, n1) = adj(Quu) applied to rows 6, 7 of the item's column of Qhat
    //    items 0..20 P[i][cc], i <= cc (stored twice);  21..56 p[i][c] (stored transposed);  the items (0, c) carry the gains of their column
    p.c_x6 = oZ; p.c_x7 = oZ; p.c_q6 = oZ; p.c_q7 = oZ; p.c_base = oZ; p.c_sg = 0; p.c_d1 = oD; p.c_d2 = oD; p.c_bd = -1; p.c_rv = RS_PAD;
    p.c_rk0 = RS_PAD; p.c_rk1 = RS_PAD;
    p.c_dump = oD4;
    if (lane < 57) {
        int r, i, cc;
        if (lane < 21) { r = 0; pair_of(lane, i, cc); } else { r = 1; i = (lane - 21) / 6; cc = (lane - 21) % 6; }
        const int qc = r ? cc + 8 : cc;
        p.c_x6 = ric_qsrc(oQ, oSG, i, 6, 1, p.c_sg); p.c_x7 = ric_qsrc(oQ, oSG, i, 7, 2, p.c_sg);
        p.c_q6 = ric_qsrc(oQ, oSG, 6, qc, 4, p.c_sg); p.c_q7 = ric_qsrc(oQ, oSG, 7, qc, 8, p.c_sg); p.c_base = ric_qsrc(oQ, oSG, i, qc, 16, p.c_sg);
        if (r) { p.c_d1 = opn + cc * 6 + i; if (i < 4) p.c_rv = RS_PV + i * 6 + cc; if (i == 0) { p.c_rk0 = RS_KF + cc; p.c_rk1 = RS_KF + OB_NC + cc; p.c_bd = 2 * cc; } }
        else {
            p.c_d1 = oPn + i * 6 + cc; if (i != cc) p.c_d2 = oPn + cc * 6 + i;
            if (i < 4) p.c_rv = RS_PX + i * 6 + cc;                              // rows 0..3 of P go to HBM (the costate recovery reads them)
            // (round 6: the mirror entry (cc, i) of rows 0..3 is no longer stored a second time -- the one reader, direction_main's costate, takes (min, max) -- one store per lane and stage less)
            if (i == 0) { p.c_rk0 = RS_K + cc; p.c_rk1 = RS_K + 6 + cc; }
        }
    }
}
// One phase-A / phase-B item: i1 + i2 + a . b with a, b four contiguous, 16-byte aligned doubles each (two ds_read_b128), i1, i2 one double each.  All six reads are issued
// and have arrived before the first fma: the compiler otherwise issues them in two groups with the first two fma between them (ISA of round 6: reads, wait, fma, fma, reads,
// wait, ...) and a phase pays two LDS round trips instead of one.  The `asm` is a scheduling fence on the six loaded values, no code.
OBCA_FN double ric_item_value(const double *pa, const double *pb, const double *pi, const double *pj) {
#ifdef OBCA_EMU
    return fma(pa[1], pb[1], fma(pa[0], pb[0], *pi)) + fma(pa[3], pb[3], fma(pa[2], pb[2], *pj));
#else
    typedef double v2d __attribute__((ext_vector_type(2)));      // (a native vector: one 128-bit register operand of the fence)
    const v2d *a2 = (const v2d *)__builtin_assume_aligned(pa, 16), *b2 = (const v2d *)__builtin_assume_aligned(pb, 16);
    v2d a0 = a2[0], a1 = a2[1], b0 = b2[0], b1 = b2[1]; double i1 = *pi, i2 = *pj;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(i1), "+v"(i2));
    return fma(a0.y, b0.y, fma(a0.x, b0.x, i1)) + fma(a1.y, b1.y, fma(a1.x, b1.x, i2));
#endif
}
// six contiguous, 16-byte aligned doubles from LDS: three ds_read_b128
OBCA_FN void ld6(const double *q, double (&v)[6]) {
#ifdef OBCA_EMU
    for (int i = 0; i < 6; i++) v[i] = q[i];
#else
    const double2 *q2 = (const double2 *)__builtin_assume_aligned(q, 16);
    const double2 a = q2[0], b = q2[1], c = q2[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
#endif
}
#ifdef OBCA_EMU
static int g_emu_ric_fail_stage = -1;      // host trace (OBCA_EMU_TRACE): the stage at which the last failed sweep met its first wrong pivot
#endif
template <int PIPE>
OBCA_FN int riccati_stage(const Inst &I, Shared &sh, const int k, const UnpackPlan (&plan)[OBCA_NLT], const RicItem (&rp)[OBCA_NLT],
                          double (&nv)[OBCA_NLT][RIC_D], const int slot, double *sg0, const gdbl *const as_base, gdbl *const rs_base, const int (&bdo)[OBCA_NLT][6]) {
    double *L = (double *)&sh;
    const int sgo = (k & 1) * OB_STG;         // which of the two stage buffers holds stage k
    // In every phase all LDS reads are issued before the first LDS write of the phase (a write may alias a later read as far as the compiler
    // knows; reads that follow a write would wait for their own round trip).
    PAR(lane) {   // phase A
        const RicItem &p = rp[LI(lane)];
        L[p.a_d] = ric_item_value(L + p.a_a + ((p.a_sg & 1) ? sgo : 0), L + p.a_b + ((p.a_sg & 2) ? sgo : 0), L + p.a_i + ((p.a_sg & 4) ? sgo : 0), L + p.a_j);
    }
    LDS_SYNC();
    double vB[OBCA_NLT];
    PAR(lane) {   // phase B
        const RicItem &p = rp[LI(lane)];
        vB[LI(lane)] = ric_item_value(L + p.b_a + ((p.b_sg & 1) ? sgo : 0), L + p.b_b, L + p.b_i + ((p.b_sg & 4) ? sgo : 0), L + p.b_j);
        L[p.b_d] = vB[LI(lane)];
        L[bdo[LI(lane)][2] + k * bdo[LI(lane)][3]] = vB[LI(lane)];      // border data of the stage: the lanes that formed q00, q10, q11 leave them there themselves (the others: dump slot)
    }
    // Quu = [q00 q10; q10 q11] must be positive definite (q00 > 0, det > 0).  Its inverse
    // is adj(Quu) / det: ONE division.  The three entries come straight out of the
    // registers of the lanes that formed them (phase-B items (delta, delta), (a, delta),
    // (a, a) = lanes 28, 40, 41), so that the pivot test and the division run in the
    // shadow of phase B's LDS round trip instead of behind it
    const double q00 = WV_READLANE(vB, 28), q10 = WV_READLANE(vB, 40), q11 = WV_READLANE(vB, 41);
    LDS_SYNC();
    PROF_FINE(I, PF_RIC_P1);
    const double det = q00 * q11 - q10 * q10;
    const int ok = UNIFORM((q00 > 0) && (det > 0) ? 1 : 0);        // (no early exit; after a failed pivot the rest of the group runs on garbage)
#ifdef OBCA_EMU
    if (!ok && g_emu_ric_fail_stage < 0) g_emu_ric_fail_stage = k;
#endif
    const double idet = rcp_nr(det);
    gdbl *ro = rs_base + (size_t)k * OB_RS;      // (as_base, rs_base: the instance's record buffers, read from LDS once per sweep -- not once per stage behind the stage's own LDS writes)
    PAR(lane) {   // phase C
        const RicItem &p = rp[LI(lane)];
        const double x6 = L[p.c_x6 + ((p.c_sg & 1) ? sgo : 0)], x7 = L[p.c_x7 + ((p.c_sg & 2) ? sgo : 0)], q6 = L[p.c_q6 + ((p.c_sg & 4) ? sgo : 0)],
                     q7 = L[p.c_q7 + ((p.c_sg & 8) ? sgo : 0)], ba = L[p.c_base + ((p.c_sg & 16) ? sgo : 0)];
        const double n0 = fma(q10, q7, -(q11 * q6)), n1 = fma(q10, q6, -(q00 * q7));       // det * gains of this column
        const double v = fma(fma(x6, n0, x7 * n1), idet, ba);
        if (PIPE) {
            const int kp = k > 0 ? k - 1 : 0, kl = k - 1 - RIC_D > 0 ? k - 1 - RIC_D : 0;
            stage_unpack_store(sg0 + (kp & 1) * OB_STG, plan[LI(lane)], nv[LI(lane)][slot]);
            stage_unpack_load(as_base + (size_t)kl * OB_AS, plan[LI(lane)], nv[LI(lane)][slot]);
        }
        L[p.c_d1] = v; L[p.c_d2] = v;
        // border data of the stage (RIC_BD doubles at the start of the dynamic block: the trajectory is dead during the sweep), without a branch around the stores and without
        // per-stage address selection: every lane stores to offset + k x stride of its own (bdo, built once per sweep); the lanes without an entry have stride 0 and the write-only slot dump4
        { double *b1 = L + bdo[LI(lane)][0] + k * bdo[LI(lane)][1]; b1[0] = q6; b1[1] = q7; L[bdo[LI(lane)][4] + k * bdo[LI(lane)][5]] = idet; }
        ro[p.c_rv] = v; ro[p.c_rk0] = (double)(n0 * idet); ro[p.c_rk1] = (double)(n1 * idet);
    }
    LDS_SYNC();
    PROF_FINE(I, PF_RIC_P2);
    return ok;
}

// the table in LDS (Shared::rit): written once per solve, read back by every sweep
#define RIC_ITEM_FIELDS(X) X(a_a) X(a_b) X(a_i) X(a_j) X(a_d) X(a_sg) X(b_a) X(b_b) X(b_i) X(b_j) X(b_d) X(b_sg) \
                           X(c_x6) X(c_x7) X(c_q6) X(c_q7) X(c_base) X(c_sg) X(c_d1) X(c_d2) X(c_bd) X(c_rv) X(c_rk0) X(c_rk1) X(c_dump)
OBCA_FN void init_ric_table(Shared &sh) {      // after Consts::N is set (the stage buffers sit behind N x 16 doubles of border data)
    PAR(lane) {
        RicItem p; ric_item(sh, lane, p); int f = 0;
#define X(n) sh.rit[f++][lane] = (ob_rit_t)p.n;
        RIC_ITEM_FIELDS(X)
#undef X
        static_assert(sizeof(RicItem) == OB_RIT_FIELDS * sizeof(int), "RIC_ITEM_FIELDS must list every field of RicItem");
    }
}
OBCA_FN void ric_item_cached(const Shared &sh, int lane, RicItem &p) {
    int f = 0;
#define X(n) p.n = sh.rit[f++][lane];
    RIC_ITEM_FIELDS(X)
#undef X
}
template <int SOC = 0>      // SOC = 1: the terminal row enters with c_soc
OBCA_FN int riccati_body(const Inst &I, Shared &sh, double rho) {   // all lanes
    const Consts &c = sh.c; const Lay &l = sh.l; const int N = UNIFORM(c.N);
    const gdbl *z = I.z; const gdbl *const as_base = I.as; gdbl *const rs_base = I.rs;
    double nv[OBCA_NLT][RIC_D];   // software pipeline, RIC_D stages deep; the slot of a stage is fixed by the unrolled loop below
    double *sg0 = ric_sg0(sh);       // the two stage buffers (dynamic LDS; in front of them the per-stage border data, behind them the operands)
    RicLds &rl = ric_lds(sh);
    UnpackPlan plan[OBCA_NLT]; RicItem rp[OBCA_NLT];
    PAR(lane) {   // terminal cost-to-go
        stage_unpack_plan(sh, lane, plan[LI(lane)]); ric_item_cached(sh, lane, rp[LI(lane)]);
        stage_unpack_constants(sh, sg0, lane);
        if (lane == 0) { rl.zero = 0.0; rl.dump = 0.0; }
        if (lane < 6) rl.zero6[lane] = 0.0;
        if (lane < 24) rl.sB[lane] = 0.0;                      // the static parts of the bilinear constants accumulate here
        const gdbl *rec = I.as + (size_t)N * OB_AS;
        if (lane < 36) {
            int i = lane / 6, j = lane % 6;
            double v = as_h(i, j) >= 0 ? rec[AS_H + as_h(i, j)] : 0.0;
            if (i == j && i < 4) v += rho;
            rl.Pn[lane] = v;
        }
        if (lane < 6) {
            double e = lane < 4 ? (SOC ? -(double)sh.soc.csoc[(l.nu - l.pi) + lane] : -(z[l.x + 4 * N + lane] - c.xF[lane])) : 0.0;
            rl.pn[0 * 6 + lane] = rec[AS_HB + lane] - (lane < 4 ? rho * e : 0.0);      // (p is kept transposed: pn[c * 6 + a])
            rl.pn[1 * 6 + lane] = lane >= 2 ? rec[AS_HT + lane - 2] : 0.0;
            for (int cc = 0; cc < 4; cc++) rl.pn[(2 + cc) * 6 + lane] = (lane == cc) ? 1.0 : 0.0;
        }
    }
    int bdo[OBCA_NLT][6];      // border-data slots of the lane: (offset, stride per stage) of its (q6, q7) pair, of its Quu entry (lanes 28, 40, 41), of 1 / det (lane 0)
    PAR(lane) {
        const RicItem &p = rp[LI(lane)]; const int oTr = (int)(g_traj - (double *)&sh), own = lane == 28 ? 12 : (lane == 40 ? 13 : (lane == 41 ? 14 : -1));
        int (&b)[6] = bdo[LI(lane)];
        b[0] = p.c_bd >= 0 ? oTr + p.c_bd : p.c_dump; b[1] = p.c_bd >= 0 ? RIC_BD : 0;
        b[2] = own >= 0 ? oTr + own : p.c_dump;        b[3] = own >= 0 ? RIC_BD : 0;
        b[4] = lane == 0 ? oTr + 15 : p.c_dump;        b[5] = lane == 0 ? RIC_BD : 0;
    }
    LDS_SYNC();
#ifdef OBCA_EMU
    g_emu_ric_fail_stage = -1;
#endif
    // head: N mod RIC_D stages with synchronous gathers, so that the pipelined loop below runs whole groups of RIC_D stages
    int k = N - 1, ok = 1;
    for (; k >= 0 && (k + 1) % RIC_D != 0 && ok; k--) {
        PAR(lane) { double v; stage_unpack_load(as_base + (size_t)k * OB_AS, plan[LI(lane)], v); stage_unpack_store(sg0 + (k & 1) * OB_STG, plan[LI(lane)], v); }
        LDS_SYNC();
        ok = riccati_stage<0>(I, sh, k, plan, rp, nv, 0, sg0, as_base, rs_base, bdo);
    }
    if (ok && k >= 0) {
        PAR(lane) {   // unpack stage k; start the gathers of stages k-1 .. k-RIC_D; enter the loop with nothing in flight
            double v; stage_unpack_load(as_base + (size_t)k * OB_AS, plan[LI(lane)], v);
            stage_unpack_store(sg0 + (k & 1) * OB_STG, plan[LI(lane)], v);
#pragma unroll
            for (int j = 0; j < RIC_D; j++) { const int st = k - 1 - j > 0 ? k - 1 - j : 0; stage_unpack_load(as_base + (size_t)st * OB_AS, plan[LI(lane)], nv[LI(lane)][(j + 1) % RIC_D]); }
#ifndef OBCA_EMU
#pragma unroll
            for (int j = 0; j < RIC_D; j++) asm volatile("" : "+v"(nv[0][j]));
#endif
        }
        LDS_SYNC();
        for (int kb = k; kb >= RIC_D - 1 && ok; kb -= RIC_D) {
#pragma unroll
            for (int ju = 0; ju < RIC_D; ju++) ok &= riccati_stage<1>(I, sh, kb - ju, plan, rp, nv, (ju + 1) % RIC_D, sg0, as_base, rs_base, bdo);
        }
    }
    if (!ok) { PROF(I, PF_RIC_BWD); return 0; }
    // bilinear constants: B(a,b) = sum over the stages of Qhat_u(a) . (adj(Quu) Qhat_u(b))
    // / det + the accumulated static parts; lane 3 p + q sums every third stage of pair p
    PAR(lane) {
        if (lane < 63) {
            int a_, b_; pair_of(lane / 3, a_, b_);
            const double *bd = g_traj;
            double acc = 0;
            for (int kk = lane % 3; kk < N; kk += 3) {
                const double *r = bd + (size_t)kk * RIC_BD;
                const double q6a = r[2 * a_], q7a = r[2 * a_ + 1], q6b = r[2 * b_], q7b = r[2 * b_ + 1], q00 = r[12], q10 = r[13], q11 = r[14], idet = r[15];
                const double n0 = fma(q10, q7b, -(q11 * q6b)), n1 = fma(q10, q6b, -(q00 * q7b));
                acc = fma(fma(q6a, n0, q7a * n1), idet, acc);
            }
            rl.TT[lane] = acc;
        }
    }
    LDS_SYNC();
    PAR(lane) {
        if (lane < 21) {
            int a_, b_; pair_of(lane, a_, b_);
            double v = (rl.TT[3 * lane] + rl.TT[3 * lane + 1]) + rl.TT[3 * lane + 2];
            // off_a . (P off_b + p_b): u1[a][b], = u2[a][b] for the right-hand sides b >= 2
            if (a_ < 2) v += b_ < 2 ? rl.sB[a_ * 6 + b_] : rl.sB[12 + a_ * 6 + b_];
            if (b_ < 2) v += rl.sB[12 + b_ * 6 + a_];                                     // off_b . p_a
            sh.Bm[a_ * 6 + b_] = v; sh.Bm[b_ * 6 + a_] = v;
        }
    }
    LDS_SYNC();
    PROF(I, PF_RIC_BWD);
    return 1;
}

// ---------------------------------------------------------------- wave-level matrix-core helpers (used by the quadcopter sweep, obca_quad_solver.h)
//   lane = 16 g + j.  wv_mfma(C, a, b): C[i][n] += sum_{k<4} a(lane (k, i)) * b(lane (k, n)); the f64 accumulator layout is register r of lane (g, j)
//   = C[g + 4r][j] (checked on the hardware by tools/micro/mfma_f64_layout.hip).  The parking blocks (6 x 14, 8 x 14) fill a third of a tile and were
//   measured slower on the matrix cores than in the three-phase LDS sweep above (round 2), so the parking sweep does not use them.
#ifdef OBCA_EMU
OBCA_FN void wv_mfma(double (&acc)[4][OBCA_NLT], const double (&a)[OBCA_NLT], const double (&b)[OBCA_NLT]) {
    double out[4][64];
    for (int l = 0; l < 64; l++) { const int g = l >> 4, n = l & 15;
        for (int r = 0; r < 4; r++) { const int i = g + 4 * r; double s_ = acc[r][l]; for (int k = 0; k < 4; k++) s_ = fma(a[16 * k + i], b[16 * k + n], s_); out[r][l] = s_; } }
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) acc[r][l] = out[r][l];
}
// lane (grp, j) -> every lane (g, j)
OBCA_FN void wv_shfl_group(double (&out)[OBCA_NLT], const double (&in)[OBCA_NLT], int grp) { for (int l = 0; l < 64; l++) out[l] = in[16 * grp + (l & 15)]; }
OBCA_FN void wv_shfl_xor(double (&out)[OBCA_NLT], const double (&in)[OBCA_NLT], int m) { for (int l = 0; l < 64; l++) out[l] = in[l ^ m]; }
#else
typedef double v4d_t __attribute__((ext_vector_type(4)));
OBCA_FN void wv_mfma(double (&acc)[4][1], const double (&a)[1], const double (&b)[1]) {
    v4d_t c = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c, 0, 0, 0);
    acc[0][0] = c[0]; acc[1][0] = c[1]; acc[2][0] = c[2]; acc[3][0] = c[3];
}
OBCA_FN void wv_shfl_group(double (&out)[1], const double (&in)[1], int grp) { out[0] = __shfl(in[0], 16 * grp + ((int)threadIdx.x & 15), 64); }
OBCA_FN void wv_shfl_xor(double (&out)[1], const double (&in)[1], int m) { out[0] = __shfl_xor(in[0], m, 64); }
#endif

template <int SOC = 0>
OBCA_FN int riccati_backward(const Inst &I, Shared &sh, double rho) {
    const int ok = riccati_body<SOC>(I, sh, rho);
    PAR(lane) { if (lane == 0) sh.ric_ok = ok; }
    SYNC();
    return sh.ric_ok;
}

