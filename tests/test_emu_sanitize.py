"""The solver source under three checking builds of the host emulation (tests/emu/obca_emu.cpp); no GPU needed.

  race   -DOBCA_EMU_RACE: every load and store of a per-instance HBM buffer is logged with the lane that issued it.  The lanes of an instance hand data to each other through
         HBM (stage records, Riccati records, the direction); unlike LDS traffic, global loads and stores of one wavefront are not ordered against each other, so a word that one
         lane stores and another lane loads or stores must be separated by a point at which the wavefront has waited for its stores (SYNC / VM_DRAIN; LDS_SYNC is not one).
         The build reports every word for which that does not hold.  DESIGN.md section 3 claims there is none (the results do not depend on timing); this is the check.
  asan   -DOBCA_EMU_ASAN under the host compiler's AddressSanitizer (flags: tests/emu_sanitize_variants.py): the per-instance buffers have exactly the sizes the HIP host code gives them (obca_hip.hip: batch_create) and the dynamic LDS block
         ends where the launch's does -- an access one double beyond any of them aborts the run.

  ubsan  the host compiler's UndefinedBehaviorSanitizer with strict bounds: an index beyond a member array of the LDS structs (the filter, the reduction scratch, the unpack tables ...), a signed overflow, a bad
         shift aborts the run.

All run full solves: uniform and ragged obstacle sets, both option sets (the second-order correction and the least-squares multipliers have phases of their own), the
minimum-distance formulation, and the quadcopter solver."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = dict(max_soc=4, recalc_y=1, lsq_init=1, restoration=1)

# the full solves both builds run; executed in a child process (the ASan runtime has to be loaded first, and a report of the race build is per process)
CHILD = r'''
import sys, os, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import emu_solver as E
from obca_amd import scenarios as S
variant = %(variant)r
lib = E.load(variant)
E.load = lambda variant=None: lib
REF = dict(max_soc=4, recalc_y=1, lsq_init=1, restoration=1)
done = []

def parking(tag, bt, N, idx, **kw):
    B = len(bt["x0"]); xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]; Ts = np.broadcast_to(bt["Ts"], (B,))
    ragged = isinstance(bt["vOb"], list)
    for i in idx:
        sl = slice(i, i + 1)
        v, A, b = (bt["vOb"][i], bt["A"][i], bt["b"][i]) if ragged else (bt["vOb"], bt["A"], bt["b"])
        v_ = np.ravel(v).astype(int); Lz = E.P.layout(N, len(v_), int(v_.sum()))
        lib.emu_set_csoc_len(C.c_int(Lz["zxL"] - Lz["pi"]))                       # the size obca_hip.hip gives the c_soc buffer (for the batch's largest layout; per instance is stricter)
        o = E.parking_signed_dist_batch(bt["x0"][sl], bt["xF"][sl], N, Ts[sl], bt["L"], bt["ego"], bt["XYbounds"], v, A, b, xWS[sl, :, 0], xWS[sl, :, 1], xWS[sl, :, 2], 0, xWS[sl], bt["uWS"][sl], **kw)
        done.append((tag, i, int(o["iters"][0]), int(o["exitflag"][0])))

bt = S.make_batch(S.BACKWARDS, 8, 40)
parking("backwards N=40", bt, 40, range(3)); parking("backwards N=40, IPOPT configuration", bt, 40, range(2), **REF)
parking("backwards N=40, minimum distance", bt, 40, range(1), dist=True)
bt = S.make_batch(S.BACKWARDS, 4, 80); parking("backwards N=80", bt, 80, range(1))
bt = S.make_batch(S.BACKWARDS, 4, 13); parking("backwards N=13 (odd horizon)", bt, 13, range(2), **REF)
mx = S.make_mixed_batch(12, 24, seed=5, min_obstacles=1, max_extra=13, rows=(3, 8), max_rows=64)
parking("1-16 obstacles of up to 8 rows", mx, 24, range(6)); parking("1-16 obstacles, IPOPT configuration", mx, 24, range(3), **REF)
q = S.make_quad_batch(2, 12, seed=5)
for kw in (dict(), dict(max_soc=4, lsq_init=1, obj_scaling=1)):
    o = E.quadcopter_signed_dist_batch(q["x0"], q["xF"], 12, q["Ts"], q["R"], q["ob"], q["xWS"], q["timeWS"], **kw)
    done.append(("quadcopter N=12 %%s" %% (kw or "",), 0, int(o["iters"][0]), int(o["exitflag"][0])))
for d in done:
    print("SOLVED", *d)
if variant == "race":
    na, nd = C.c_long(0), C.c_long(0)
    n = lib.emu_race_report(C.byref(na), C.byref(nd))
    print("SUMMARY hazards %%d accesses %%d dummy_stores %%d" %% (n, na.value, nd.value))
'''


def _run(variant, env=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_solver
    emu_solver.build(variant)                                        # compile in the parent: a failing build shows as such
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, variant=variant)], capture_output=True, text=True, env=e, timeout=1500)
    return r


@pytest.mark.timeout(1800)
def test_no_cross_lane_hazard_through_hbm_between_two_drains():
    r = _run("race")
    assert r.returncode == 0, r.stderr[-3000:]
    solved = [l for l in r.stdout.splitlines() if l.startswith("SOLVED")]
    hazards = [l for l in r.stdout.splitlines() if l.startswith("HAZARD")]
    summary = [l for l in r.stdout.splitlines() if l.startswith("SUMMARY")][0].split()
    assert len(solved) >= 20 and sum(int(l.split()[-1]) == 1 for l in solved) >= len(solved) - 2, solved      # the solves are real ones (exit flag 1)
    assert int(summary[4]) > 10_000_000, summary                     # the log saw the traffic (tens of millions of accesses)
    assert int(summary[6]) > 0, summary                              # ... including the stores to the dummy slot of the Riccati record, the one shared word by design
    assert not hazards, "\n".join(hazards)


@pytest.mark.timeout(1800)
def test_no_access_beyond_the_sizes_the_host_code_allocates():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no AddressSanitizer runtime next to gcc")
    r = _run("asan", env=dict(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"))
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, r.stderr[-3000:]
    solved = [l for l in r.stdout.splitlines() if l.startswith("SOLVED")]
    assert len(solved) >= 20, r.stdout[-2000:]


@pytest.mark.timeout(1800)
def test_no_undefined_behaviour_in_the_solver_source():
    ub = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(ub) or not os.path.exists(ub):
        pytest.skip("no UndefinedBehaviorSanitizer runtime next to gcc")
    r = _run("ubsan", env=dict(LD_PRELOAD=ub, UBSAN_OPTIONS="print_stacktrace=1"))
    assert "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.startswith("SOLVED")]) >= 20, r.stdout[-2000:]
