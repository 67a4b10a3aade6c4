"""PCIe-inclusive rate of the host-pointer entry point: obca_parking_signed_dist_batch from host arrays to host arrays (packing into pinned
staging, H2D, DualMultWS + IPM kernels, D2H of the outputs, unpacking; chunks pipelined over the context's worker lanes), config-2 batches."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
N = 80
res = {}
for B in [int(a) for a in (sys.argv[1:] or ["1024", "4096"])]:
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    ts, tc, tr, trc = [], [], [], []
    rx, ry, ryaw = (np.ascontiguousarray(xWS[:, :, q]) for q in range(3))
    for rep in range(5):       # fresh output arrays every call (first-touch page faults inside the C call)
        t0 = time.perf_counter()
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], rx, ry, ryaw, 0, xWS, bt["uWS"])
        ts.append(time.perf_counter() - t0); tc.append(out["time"])
    keep = {}
    for rep in range(5):       # the caller keeps its output arrays between calls
        t0 = time.perf_counter()
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], rx, ry, ryaw, 0, xWS, bt["uWS"], buffers=keep)
        tr.append(time.perf_counter() - t0); trc.append(out["time"])
    ctx = OA.Context(0)
    b = OA.Batch(ctx, B, N)
    t0 = time.perf_counter()
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    t1 = time.perf_counter(); b.solve(); t2 = time.perf_counter(); o2 = b.download(); t3 = time.perf_counter()
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    t4 = time.perf_counter(); b.solve(); t5 = time.perf_counter(); o2 = b.download(); t6 = time.perf_counter()
    b.close(); ctx.close()
    res[B] = dict(one_shot_ms=[round(1e3 * t, 2) for t in ts], one_shot_solves_per_s=round(B / min(ts[1:]), 1), c_call_ms=[round(1e3 * t, 2) for t in tc],
                  reused_outputs_ms=[round(1e3 * t, 2) for t in tr], reused_outputs_c_call_ms=[round(1e3 * t, 2) for t in trc], reused_outputs_solves_per_s=round(B / min(tr[1:]), 1), resident_upload_ms=round(1e3 * (t4 - t3), 2),
                  resident_solve_ms=round(1e3 * (t5 - t4), 2), resident_download_ms=round(1e3 * (t6 - t5), 2), converged=int((out["exitflag"] == 1).sum()),
                  chunk=os.environ.get("OBCA_CHUNK", "default"), slots=os.environ.get("OBCA_SLOTS", "default"))
print(json.dumps(res))
