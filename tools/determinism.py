"""solve the config-2 batch repeatedly and compare the results bit for bit (race detector)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
from obca_amd import api as _api
_lib = _api._load()
if not hasattr(_lib, 'obca_batch_set_formulation'):
    _lib.obca_batch_set_formulation = lambda *a: 0        # older builds (bisecting)
B, N = 1024, 80; R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
b = OA.Batch(OA.Context(0), B, N)
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
ref = None; bad = 0
for r in range(R):
    b.solve(); o = b.download()
    if ref is None: ref = o; continue
    dif = np.flatnonzero((o["iters"] != ref["iters"]) | (np.abs(o["xp"] - ref["xp"]).max(axis=(1, 2)) > 0))
    if len(dif):
        bad += 1; i = dif[0]
        print("run", r, "differs in", len(dif), "instances, first", i, "iters", ref["iters"][i], o["iters"][i], "max|dx| %.3e" % np.abs(o["xp"][i] - ref["xp"][i]).max())
print(os.environ.get("OBCA_HIP_LIBRARY", "default"), "runs", R, "nondeterministic runs", bad)
