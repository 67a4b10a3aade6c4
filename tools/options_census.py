"""Which of the three IPOPT switches moves a solve into another local solution, and what does each cost?  

The bench batches of configs 2 / 3 / 5 under all eight combinations of (max_soc = 4, recalc_y, lsq_init): kernel time of one launch, iterations, passes, and the number of
instances whose solution differs from the one the reference's configuration (all three on) finds, beyond the path's stated tolerance (states / inputs 1e-3, time scale 1e-4,
objective 1e-4 relative)."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S

cfgs = [int(a) for a in sys.argv[1:]] or [2, 5]
for cfg in cfgs:
    N = 80
    if cfg == 2:
        bt = S.make_batch(S.BACKWARDS, 1024, N)
    elif cfg == 3:
        bt = S.make_batch(S.PARALLEL, 1024, N, seed=20260925, goal_jitter=True)
    else:
        bt = S.make_mixed_batch(4096, N, seed=20260925, min_obstacles=1)
    B = len(bt["x0"])
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    res = {}
    for soc, rc, lsq in itertools.product((0, 4), (0, 1), (0, 1)):
        o = OA.default_opts(); o.max_soc = soc; o.recalc_y = rc; o.lsq_init = lsq
        b.solve(opts=o); b.solve(opts=o); ms = b.kernel_ms()[0]
        res[(soc, rc, lsq)] = (b.download(), ms)
    ref = res[(4, 1, 1)][0]
    print("config %d, %d instances; reference configuration = (max_soc 4, recalc_y 1, lsq_init 1)" % (cfg, B))
    for key, (o, ms) in res.items():
        both = (o["exitflag"] == 1) & (ref["exitflag"] == 1)
        dx = np.array([np.abs(np.asarray(o["xp"][i]) - np.asarray(ref["xp"][i])).max() for i in range(B)])
        du = np.array([np.abs(np.asarray(o["up"][i]) - np.asarray(ref["up"][i])).max() for i in range(B)])
        df = np.abs(o["obj"] - ref["obj"]) / np.maximum(1.0, np.abs(ref["obj"])); dts = np.abs(o["timeScale"][:, 0] - ref["timeScale"][:, 0])
        differs = both & ((dx > 1e-3) | (du > 1e-3) | (df > 1e-4) | (dts > 1e-4))
        print("  max_soc %d recalc_y %d lsq_init %d : one launch %.3f ms, solved %d, iterations %.2f, passes %.2f, another local solution than the reference configuration: %d"
              % (key + (ms, int((o["exitflag"] == 1).sum()), o["info"][:, 1].mean(), (o["info"][:, 1] + o["info"][:, 6]).mean(), int(differs.sum()))), flush=True)
    b.close(); ctx.close()
