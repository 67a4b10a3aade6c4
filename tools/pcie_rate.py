"""PCIe-inclusive rate of the host-pointer entry point: obca_parking_signed_dist_batch from host arrays to host arrays (allocation, packing,
H2D, DualMultWS + IPM kernels, D2H, unpacking), config-2 batch."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
B, N = 1024, 80
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"],
                                       xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    ts.append(time.perf_counter() - t0)
b = OA.Batch(OA.Context(0), B, N)
t0 = time.perf_counter()
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
t1 = time.perf_counter(); b.solve(); t2 = time.perf_counter(); o2 = b.download(); t3 = time.perf_counter()
print(json.dumps(dict(B=B, one_shot_ms=[round(1e3 * t, 2) for t in ts], one_shot_solves_per_s=B / min(ts[1:]), upload_ms=1e3 * (t1 - t0), solve_ms=1e3 * (t2 - t1),
                      download_ms=1e3 * (t3 - t2), converged=int((out["exitflag"] == 1).sum()))))
