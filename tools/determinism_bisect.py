"""Where do two runs of the same inputs part ways when the GPU is shared?  (round 5: run next to heavy co-runners; the job script is in the git history, docs/HISTORY.md "Round 5 detail")
  1. DualMultWS alone (obca_dualmult_ws_batch), repeated: lam / mu / d bit for bit;
  2. the interior point cut off after K factorisation passes (OBCA_SLICE_PASSES=K, OBCA_SLICE_ONLY=1: every instance parks with its iterate in place), repeated, for K = 1, 2, 4, 8, 16:
     which of x, u, t, lambda, mu, sl / info differ, on how many instances."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S
N, B, R = 80, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 12
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
ref = None; bad = 0
for r in range(R):
    ls, ns, ds = OA.dualmult_ws_batch(N, bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], bt["ego"])
    cur = (np.asarray(ls), np.asarray(ns), np.asarray(ds))
    if ref is None: ref = cur; continue
    bad += int(any(not np.array_equal(a, b) for a, b in zip(cur, ref)))
print("DualMultWS alone: %d runs, %d differ from the first" % (R, bad), flush=True)
b = OA.Batch(OA.Context(0), B, N)
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
os.environ["OBCA_SLICE_ONLY"] = "1"
for K in (1, 2, 4, 8, 16):
    os.environ["OBCA_SLICE_PASSES"] = str(K)
    ref = None; cnt = {}; runs_bad = 0
    for r in range(R):
        b.solve(); o = b.download()
        if ref is None: ref = o; continue
        anyd = False
        for k in ("xp", "up", "timeScale", "info"):
            d = (np.asarray(o[k]) != np.asarray(ref[k])).reshape(B, -1).any(axis=1)
            cnt[k] = cnt.get(k, 0) + int(d.sum()); anyd |= bool(d.any())
        for k in ("lp", "np", "sl"):
            d = np.array([not np.array_equal(o[k][i], ref[k][i]) for i in range(B)])
            cnt[k] = cnt.get(k, 0) + int(d.sum()); anyd |= bool(d.any())
        runs_bad += anyd
    print("interior point stopped after %2d passes: %d of %d runs differ from the first; differing instances per field (summed over the runs): %s" % (K, runs_bad, R - 1, cnt), flush=True)
