"""quadcopter path on the GPU: batch timing at the shipped scenario size (N=60) with jittered start / goal."""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd
from obca_amd import scenarios as S
from obca_amd.api import QuadBatch, _ctx

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bt = S.make_quad_batch(B, N)
qb = QuadBatch(_ctx(0), B, N)
qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"])
for rep in range(3):
    t0 = time.perf_counter(); qb.solve(); dt = time.perf_counter() - t0
    ms = qb.kernel_ms()
out = qb.download()
it = out["iters"]
print(json.dumps(dict(B=B, N=N, kernel_ms=ms, wall_ms=dt * 1e3, solves_per_s=B / (ms * 1e-3), converged=float((out["exitflag"] == 1).mean()),
                      flag2=float((out["exitflag"] == 2).mean()), iters_mean=float(it.mean()), iters_max=int(it.max()), nreg_mean=float(out["info"][:, 6].mean()),
                      scratch_MB=qb.scratch_bytes() / 1e6)))
pc = qb.phase_cycles()
if pc.sum() > 0:
    names = "init asm_obs asm_stage riccati border fwd_setup fwd_seq bs_stage bs_obs trial apply other".split()      # fwd_setup: gather tables, kf_k(coef) of all stages, first gathers of the forward sweep
    passes = out["info"][:, 1] + out["info"][:, 6]; tot = pc[:, :len(names)].sum(1)
    print("cycles per pass: mean %.0f" % (tot / passes).mean())
    for i, n in enumerate(names):
        print("%-12s %5.1f%%   cycles/pass %9.0f" % (n, 100 * pc[:, i].sum() / tot.sum(), (pc[:, i] / passes).mean()))
    if pc[:, 12:16].sum() > 0:       # segments of one stage of the backward sweep (pipelined part), clocks per pass
        for i, n in enumerate(("ric: operands + 20 MFMA", "ric: Quu readlane + LDL + shuffles", "ric: gains solve + stores", "ric: 3 MFMA + symmetrise P via LDS")):
            print("%-36s cycles/pass %9.0f" % (n, (pc[:, 12 + i] / passes).mean()))
