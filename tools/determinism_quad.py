"""quadcopter path: repeated solves of one batch compared bit for bit (race detector)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from obca_amd import scenarios as S
from obca_amd.api import QuadBatch, _ctx
B = 256; N = 60; R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bt = S.make_quad_batch(B, N)
qb = QuadBatch(_ctx(0), B, N)
qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"])
ref = None; bad = 0
for r in range(R):
    qb.solve(); o = qb.download()
    if ref is None: ref = o; continue
    dif = np.flatnonzero((o["iters"] != ref["iters"]) | (np.abs(o["xp"] - ref["xp"]).max(axis=(1, 2)) > 0))
    if len(dif):
        bad += 1; print("run", r, "differs in", len(dif), "instances; first", dif[0], ref["iters"][dif[0]], o["iters"][dif[0]])
print("quad runs", R, "nondeterministic runs", bad)
