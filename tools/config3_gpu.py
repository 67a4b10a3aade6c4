"""BASELINE config 3: parallel parking (4 obstacles / 6 rows), B instances with Hybrid A* warm starts planned on the host cores first."""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from obca_amd import scenarios as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; N = 80
t0 = time.time(); bt = S.make_batch(S.PARALLEL, B, N); t_plan = time.time() - t0        # forks workers: before any HIP context exists
import obca_amd as OA
b = OA.Batch(OA.Context(0), B, N)
xWS = bt["xWS"]
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
for _ in range(3):
    b.solve()
ipm, dws = b.kernel_ms(); out = b.download()
ps = out["info"][:, 1] + out["info"][:, 6]
print(json.dumps(dict(B=B, N=N, plan_s=t_plan, ipm_ms=ipm, dualws_ms=dws, solves_per_s=float((out["exitflag"] == 1).sum()) / ((ipm + dws) * 1e-3),
                      converged=float((out["exitflag"] == 1).mean()), iters_mean=float(out["iters"].mean()), iters_max=int(out["iters"].max()),
                      passes_mean=float(ps.mean()), passes_max=int(ps.max()), max_slack=float(max(s.max() for s in out["sl"])))))
