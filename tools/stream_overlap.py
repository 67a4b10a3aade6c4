"""throughput when config-2 batches are streamed: S device-resident batches on S streams, launches not synchronised per step, so the tail
of one batch (a few hard instances) overlaps the bulk of the next.  Not the bench contract's step (bench.py synchronises every step)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
B, N = 1024, 80
nS = int(sys.argv[1]) if len(sys.argv) > 1 else 2; K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bs = []
for s in range(nS):
    bt = S.make_batch(S.BACKWARDS, B, N, seed=20260925 + s)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    b = OA.Batch(OA.Context(0), B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    b.solve(); bs.append(b)
t0 = time.perf_counter()
for k in range(K):
    bs[k % nS].solve(sync=False)
for b in bs:
    b.sync()
dt = time.perf_counter() - t0
conv = sum(int((b.download()["exitflag"] == 1).sum()) for b in bs) / nS
print(json.dumps(dict(streams=nS, steps=K, ms_per_step=1e3 * dt / K, solves_per_s=conv * K / dt, last_kernel_ms=[b.kernel_ms()[0] for b in bs])))
