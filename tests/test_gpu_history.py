"""Same inputs, same bits, WHATEVER RAN ON THE GPU BEFORE (run with -m gpu; the last file of the GPU suite, tests/conftest.py).  Until the end of round 5 the parking kernels read two
LDS words they never wrote (the multiplier sums behind IPOPT's termination scaling factors: their stores had slipped into a comment).  With zeros, NaNs or the previous
parking workgroup's own leftovers there the solve is the right one; with large numbers left by ANOTHER kernel -- a foreign process on the GPU, a workgroup of another horizon whose
dynamic LDS block lay at that place -- it ends about five iterations early.  That is what the driver's round-4 run and two of the boxes leased in round 5 met.
These tests leave such patterns on purpose, through the diagnostic library libobca_diag.so (include/obca_diag.h, obca_amd/diag.py -- not part of the product library), and interleave batches of other shapes.  The same check runs on the CPU
against the emulation (tests/test_emu_cpu.py: finite poison patterns) -- registers and the GPU-only code paths can only be covered here."""
import numpy as np
import pytest
from conftest import gpu_verdict
from obca_amd import scenarios as S
from test_gpu_determinism import _args, _resident, _diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def OA():
    import obca_amd
    obca_amd.Context(0).close()
    return obca_amd


PATTERNS = [(1e30, "large"), (-1e30, "large_negative"), (0.5, "small"), (float("nan"), "nan")]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("value", [p[0] for p in PATTERNS], ids=[p[1] for p in PATTERNS])
def test_results_do_not_depend_on_what_other_kernels_left_on_the_compute_units(OA, value):
    """obca_diag_leave_pattern (libobca_diag.so) fills the LDS of every CU (mask 4), then registers + LDS + scratch (mask 15), with `value`; the solves that follow return the bits of the solve
    before it.  (Until the end of round 5 a large pattern ended the parking solves ~5 iterations early: DESIGN.md section 11.  NaN alone never showed it: it passes through fmax.)"""
    from obca_amd import diag
    for N, B, optset in ((40, 1024, "default"), (80, 512, "ipopt")):
        bt = S.make_batch(S.BACKWARDS, B, N)
        opts = OA.ipopt_opts() if optset == "ipopt" else None
        _, xWS = _args(bt, N)
        ctx = OA.Context(0)
        b = OA.Batch(ctx, B, N)
        b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
        b.solve(opts=opts); ref = b.download()
        assert (ref["exitflag"] == 1).mean() > 0.95
        for mask in (4, 15):
            reach = diag.leave_pattern(ctx, mask, value)
            assert reach[0][1] >= 0.9 * reach[0][0] > 0, reach      # the pattern kernel reached the machine: (nearly) every compute unit ran the four workgroups that cover its LDS
            b.solve(opts=opts); d = _diff(b.download(), ref)
            assert d == "", "N %d, %s options, pattern %r in %s: %s" % (N, optset, value, "LDS" if mask == 4 else "registers, LDS and scratch", d) + gpu_verdict()
        b.close(); ctx.close()
    N, B = 20, 256
    q = S.make_quad_batch(B, N, seed=5)
    ctx = OA.Context(0)
    qb = OA.QuadBatch(ctx, B, N); qb.upload(q["x0"], q["xF"], q["Ts"], q["R"], q["ob"], q["xWS"], q["timeWS"])
    qb.solve(); ref = qb.download()
    diag.leave_pattern(ctx, 15, value)
    qb.solve(); o = qb.download()
    for k in ("xp", "up", "timeScale", "exitflag", "lp", "slack", "info"):
        assert np.array_equal(np.asarray(o[k]), np.asarray(ref[k])), ("quadcopter", value, k)
    qb.close(); ctx.close()


@pytest.mark.timeout(600)
def test_results_do_not_depend_on_the_batches_solved_in_between(OA):
    """A batch keeps its bits when batches of other horizons (their workgroups' dynamic LDS blocks lie where this batch's static block will) and of the quadcopter solver have run
    on the GPU since its last solve."""
    N, B = 80, 1024
    bt = S.make_batch(S.BACKWARDS, B, N)
    ref = _resident(OA, bt, N, None, repeats=1)[0]
    refi = _resident(OA, bt, N, OA.ipopt_opts(), repeats=1)[0]
    others = [(S.make_batch(S.BACKWARDS, 700, 24), 24), (S.make_mixed_batch(300, 40, seed=5, min_obstacles=1, max_extra=13, rows=(3, 8), max_rows=64), 40), (S.make_batch(S.BACKWARDS, 400, 120), 120)]
    q = S.make_quad_batch(256, 20, seed=5)
    for rnd in range(3):
        for ob, oN in others:
            _resident(OA, ob, oN, None if rnd % 2 else OA.ipopt_opts(), repeats=1)
            ctx = OA.Context(0)
            qb = OA.QuadBatch(ctx, 256, 20); qb.upload(q["x0"], q["xF"], q["Ts"], q["R"], q["ob"], q["xWS"], q["timeWS"]); qb.solve(); qb.download(); qb.close(); ctx.close()
            d = _diff(_resident(OA, bt, N, None, repeats=1)[0], ref)
            assert d == "", "round %d, after a batch of horizon %d: %s" % (rnd, oN, d) + gpu_verdict()
            d = _diff(_resident(OA, bt, N, OA.ipopt_opts(), repeats=1)[0], refi)
            assert d == "", "round %d, after a batch of horizon %d, IPOPT configuration: %s" % (rnd, oN, d) + gpu_verdict()
