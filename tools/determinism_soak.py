"""Soak test: does a GPU start to differ from itself under SUSTAINED load?  (Both boxes that failed bit-equality tests in round 5 did so in jobs that had kept the GPU busy for
minutes; 29 boxes probed for 3-8 s each were clean.)  The config-2 bench batch is solved over and over for `seconds` (default 240) with four copies in flight; every download is
compared with the first solve; a line per 10 s: elapsed time, solves so far, differing (instance, download) pairs in the interval, the GPU's temperature / clock / power
(rocm-smi).  Every `probe_every` seconds tools/micro/cu_consistency runs 20 launches (is the pure-pattern probe clean while the solver is not?)."""
import os, sys, time, subprocess, re
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S

T = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
OPT = sys.argv[2] if len(sys.argv) > 2 else "ipopt"
opts = OA.ipopt_opts() if OPT == "ipopt" else None
N, B = 80, 1024


def smi():
    try:
        t = subprocess.run(["rocm-smi", "--showtemp", "--showclocks", "--showpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20).stdout.decode(errors="replace")
        g = lambda pat: (re.search(pat, t) or [None, "?"])[1]
        return "junction %s C, memory %s C, sclk %s, power %s W" % (g(r"junction\) \(C\): ([\d.]+)"), g(r"memory\) \(C\): ([\d.]+)"), g(r"sclk clock level: \S+ \((\d+Mhz)\)"), g(r"Power \(W\): ([\d.]+)"))
    except Exception as e:
        return "rocm-smi: %r" % e


bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
bs = []
for _ in range(4):
    b = OA.Batch(OA.Context(0), B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    bs.append(b)
bs[0].solve(opts=opts); ref = bs[0].download()
print("soak: %d-instance batch, options %s, %.0f s; start: %s" % (B, OPT, T, smi()), flush=True)
t0 = time.time(); tl = t0; solves = 0; bad_iv = 0; bad_tot = 0; dl = 0; tp = t0
here = os.path.dirname(os.path.abspath(__file__))
while time.time() - t0 < T:
    for k in range(16):                       # 16 launches in flight over the four copies, then one download per copy
        bs[k % 4].solve(opts=opts, sync=False)
    for b in bs:
        b.sync(); o = b.download(); dl += 1
        dif = np.flatnonzero((o["info"] != ref["info"]).any(axis=1) | (np.abs(o["xp"] - ref["xp"]).reshape(B, -1).max(axis=1) > 0))
        if len(dif):
            bad_iv += len(dif)
            if bad_tot + bad_iv <= 40:
                print("   t=%.0fs download %d: %d instances differ, e.g. %s" % (time.time() - t0, dl, len(dif), [(int(i), int(o["iters"][i]), int(ref["iters"][i])) for i in dif[:6]]), flush=True)
    solves += 16 * B
    if time.time() - tl >= 10:
        bad_tot += bad_iv
        print("t=%4.0fs  solves %8d  differing (instance, download) pairs in this interval %4d (total %d of %d downloads x %d)  %s" % (time.time() - t0, solves, bad_iv, bad_tot, dl, B, smi()), flush=True)
        bad_iv = 0; tl = time.time()
    if time.time() - tp >= 60:
        try:
            r = subprocess.run([os.path.join(here, "micro", "cu_consistency"), "20"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120).stdout.decode(errors="replace").strip().splitlines()
            print("   cu_consistency:", r[-1] if r else "no output", flush=True)
        except Exception as e:
            print("   cu_consistency failed:", e)
        tp = time.time()
print("soak done: %d solves, %d differing (instance, download) pairs in %d downloads; end: %s" % (solves, bad_tot + bad_iv, dl, smi()))
