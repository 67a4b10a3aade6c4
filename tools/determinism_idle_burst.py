"""Do results differ right after the GPU wakes up?  (Both non-reproducing boxes of round 5 showed up in pytest runs -- GPU bursts between seconds of CPU-only oracle work, with 64
oracle processes on 16 CPUs -- and none of 60 short probes or a 240 s soak did.)  For `seconds`: sleep 0.3-5 s (the GPU drops to its low-power state), then three solves of the
bench batch back to back, each compared with the reference bit for bit; the second half of the run adds CPU load (busy processes) during the idle phases.  Reports differing
(instance, solve) pairs by position after the wake-up."""
import os, sys, time, subprocess, multiprocessing as mp
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S

T = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
N, B = 80, 1024


def burn(stop):
    x = 1.0
    while not stop.is_set():
        for _ in range(200000):
            x = x * 1.0000001 + 1e-9


def main():
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    b = OA.Batch(OA.Context(0), B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    opts = OA.ipopt_opts()
    b.solve(opts=opts); ref = b.download()
    rng = np.random.default_rng(5)
    bad = [0, 0, 0]; n = 0; t0 = time.time(); stop = None; procs = []
    while time.time() - t0 < T:
        if time.time() - t0 > T / 2 and not procs:      # second half: CPU load as a pytest run's oracle pools give
            stop = mp.Event(); procs = [mp.Process(target=burn, args=(stop,)) for _ in range(48)]
            [p.start() for p in procs]
            print("t=%.0fs: 48 busy CPU processes started" % (time.time() - t0), flush=True)
        time.sleep(float(rng.uniform(0.3, 5.0)))
        for k in range(3):
            b.solve(opts=opts); o = b.download()
            d = int(((o["info"] != ref["info"]).any(axis=1) | (np.abs(o["xp"] - ref["xp"]).reshape(B, -1).max(axis=1) > 0)).sum())
            bad[k] += d
            if d:
                print("   t=%.0fs wake-up %d, solve %d after it: %d instances differ" % (time.time() - t0, n, k, d), flush=True)
        n += 1
    if procs:
        stop.set(); [p.join(timeout=10) for p in procs]
    print("idle/burst: %d wake-ups in %.0f s; differing (instance, solve) pairs by position after the wake-up: first %d, second %d, third %d" % (n, time.time() - t0, bad[0], bad[1], bad[2]))


if __name__ == "__main__":
    main()
