"""BASELINE config 5 demo: B instances with 3..10 obstacles of 1..4 rows each in one launch (irregular H-rep packing)."""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; N = 80
bt = S.make_mixed_batch(B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
b = OA.Batch(OA.Context(0), B, N)
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
for _ in range(2):
    b.solve()
ipm, dws = b.kernel_ms(); out = b.download()
nob = np.array([len(v) for v in bt["vOb"]])
print(json.dumps(dict(B=B, N=N, ipm_ms=ipm, dualws_ms=dws, solves_per_s=B / ((ipm + dws) * 1e-3), converged=float((out["exitflag"] == 1).mean()),
                      iters_mean=float(out["iters"].mean()), iters_max=int(out["iters"].max()), nOb_mean=float(nob.mean()),
                      M_mean=float(np.mean([v.sum() for v in bt["vOb"]])), scratch_GB=b.scratch_bytes() / 1e9)))
