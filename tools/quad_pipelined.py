"""quadcopter path with several batches in flight (one context / HIP stream each), like bench.py does for the parking path"""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd
from obca_amd import scenarios as S
from obca_amd.api import QuadBatch, Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 60; steps = 8
bt = S.make_quad_batch(B, N)
res = {}
for nS in (1, 2, 3):
    qs = []
    for _ in range(nS):
        qb = QuadBatch(Context(0), B, N)
        qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"])
        qs.append(qb)
    for q in qs:
        q.solve()
    t0 = time.perf_counter()
    for k in range(steps):
        qs[k % nS].solve(sync=False)
    for q in qs:
        q.sync()
    dt = time.perf_counter() - t0
    out = qs[0].download()
    res[nS] = dict(ms_per_step=dt / steps * 1e3, solves_per_s=float((out["exitflag"] == 1).sum()) * steps / dt)
    for q in qs:
        q.close()
print(json.dumps(dict(B=B, N=N, steps=steps, streams=res)))
