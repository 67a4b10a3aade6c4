"""Oracle A/B of an IPOPT semantic (switches = (max_soc, recalc_y, lsq_init) of the oracle options, set explicitly): exit flags and iteration counts
on the config-5 distribution (1-10 obstacles) and on config 3 (parallel parking, goal jitter).  python tools/soc_probe.py [B5] [B3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from obca_amd import scenarios as S
import oracle_pool as P


def stats(tag, res):
    fl = np.array([r[1] for r in res]); it = np.array([r[2] for r in res])
    print(f"{tag}: ok {np.sum(fl == 1)}/{len(fl)}  flags {dict(zip(*np.unique(fl, return_counts=True)))}  iters mean {it.mean():.1f} p50 {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}", flush=True)
    return fl, it


if __name__ == "__main__":
    B5 = int(sys.argv[1]) if len(sys.argv) > 1 else 512; B3 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    envs = [None, (4, 0, 0)] + ([(0, 1, 0), (4, 1, 0)] if "--recalc" in sys.argv else [])
    if "--lsq" in sys.argv: envs = [None, (0, 0, 1)]
    bt5 = S.make_mixed_batch(B5, 80, seed=20260925, min_obstacles=1) if B5 else None
    bt3 = S.make_batch(S.PARALLEL, B3, 80, seed=20260925, goal_jitter=True) if B3 else None
    out = {}
    for e in envs:
        t0 = time.time()
        if bt5: out[("c5", str(e))] = stats(f"config5 {e}", P.mixed_oracle_all(bt5, bt5["xWS"], workers=8, switches=e))
        if bt3: out[("c3", str(e))] = stats(f"config3 {e}", P.parking_oracle_all(bt3, bt3["xWS"], workers=8, switches=e))
        print("  %.0f s" % (time.time() - t0), flush=True)
    base = str(envs[0])
    for (c, e), (fl, it) in out.items():
        if e != base:
            f0, i0 = out[(c, base)]
            print(c, e, "fixed", int(np.sum((f0 != 1) & (fl == 1))), "broken", int(np.sum((f0 == 1) & (fl != 1))), "iters changed on", int(np.sum(it != i0)), "sum iters", int(i0.sum()), "->", int(it.sum()))
