"""Which hardware unit produced a result that differs between two runs?  (needs the -DOBCA_HWID build: OBCA_HIP_LIBRARY=.../variants/libobca_hip_hwid.so)

Solves the config-2 bench batch (1 024 instances, N = 80: every CU holds four instances) R times, compares every run with the first one bit for bit and lists, for the
instances that differ, the XCC / shader engine / CU / SIMD they ran on in both runs; then the same for smaller batches (which CUs do they reach?) and the device's
identification (rocm-smi: firmware, partitions, clocks, RAS counters).  A defect of the code would spread over the units; a defect of a unit stays on it."""
import os, sys, subprocess, collections
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S, api

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = api._load()


def hw(b, B):
    out = np.zeros((B, 16))
    assert lib.obca_batch_debug_phase_cycles(b._h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    h = out[:, 14].astype(np.int64); x = out[:, 15].astype(np.int64) & 0xF
    return dict(xcc=x, se=(h >> 13) & 7, sh=(h >> 12) & 1, cu=(h >> 8) & 15, simd=(h >> 4) & 3, wave=h & 15)


def unit(u, i):
    return "xcc%d se%d sh%d cu%d simd%d" % (u["xcc"][i], u["se"][i], u["sh"][i], u["cu"][i], u["simd"][i])


def run(B, N, reps, opts=None, tag=""):
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    ref = None; bad_units = collections.Counter(); all_units = collections.Counter(); nbad = 0
    for r in range(reps):
        b.solve(opts=opts); o = b.download(); u = hw(b, B)
        for i in range(B):
            all_units[(int(u["xcc"][i]), int(u["se"][i]), int(u["sh"][i]), int(u["cu"][i]))] += 1
        if ref is None:
            ref, uref = o, u; continue
        dif = np.flatnonzero((o["info"] != ref["info"]).any(axis=1) | (np.abs(o["xp"] - ref["xp"]).reshape(B, -1).max(axis=1) > 0))
        nbad += len(dif)
        for i in dif:
            bad_units[(int(u["xcc"][i]), int(u["se"][i]), int(u["sh"][i]), int(u["cu"][i]))] += 1
            bad_units[("first-run", int(uref["xcc"][i]), int(uref["se"][i]), int(uref["sh"][i]), int(uref["cu"][i]))] += 1
        if len(dif):
            print("  %s run %d: %d instances differ from run 0: %s" % (tag, r, len(dif), ", ".join("%d [%s | run0 %s] iters %d|%d" % (i, unit(u, i), unit(uref, i), o["iters"][i], ref["iters"][i]) for i in dif[:12])), flush=True)
    b.close(); ctx.close()
    print("%s B %d N %d: %d runs, %d differing (instance, run) pairs; distinct (xcc, se, sh, cu) units used %d" % (tag, B, N, reps, nbad, len(all_units)), flush=True)
    if bad_units:
        print("   units of the differing instances (this run):", sorted([(k, v) for k, v in bad_units.items() if k[0] != "first-run"], key=lambda kv: -kv[1])[:24])
        print("   units of the same instances in run 0       :", sorted([(k[1:], v) for k, v in bad_units.items() if k[0] == "first-run"], key=lambda kv: -kv[1])[:24])
    return nbad


def run_copies(B, N, rounds, opts=None, tag=""):
    """bench.py's regime: four device-resident copies of one batch on four contexts / streams, solved concurrently, every download compared with a lone solve"""
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    bs = []
    for _ in range(4):
        b = OA.Batch(OA.Context(0), B, N)
        b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
        bs.append(b)
    bs[0].solve(opts=opts); ref = bs[0].download(); nbad = 0
    for r in range(rounds):
        for k in range(8):
            bs[k % 4].solve(opts=opts, sync=False)
        for i, b in enumerate(bs):
            b.sync(); o = b.download(); u = hw(b, B)
            dif = np.flatnonzero((o["info"] != ref["info"]).any(axis=1) | (np.abs(o["xp"] - ref["xp"]).reshape(B, -1).max(axis=1) > 0))
            nbad += len(dif)
            if len(dif):
                print("  %s round %d copy %d: %d instances differ from the lone solve: %s" % (tag, r, i, len(dif), ", ".join("%d [%s] iters %d|%d" % (j, unit(u, j), o["iters"][j], ref["iters"][j]) for j in dif[:12])), flush=True)
    for b in bs:
        b.close()
    print("%s B %d N %d: 4 copies x %d rounds, %d differing (instance, download) pairs" % (tag, B, N, rounds, nbad), flush=True)
    return nbad


print(os.environ.get("OBCA_HIP_LIBRARY", "default library"))
for cmd in (["rocm-smi", "--showproductname", "--showfwinfo", "--showcomputepartition", "--showmemorypartition", "--showclocks", "--showrasinfo", "all", "--showperflevel", "--showpower", "--showtemp"],
            ["rocminfo"]):
    try:
        t = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode(errors="replace")
        if cmd[0] == "rocminfo":
            t = "\n".join(ln for ln in t.splitlines() if any(k in ln for k in ("Marketing", "Compute Unit", "Max Clock", "gfx", "Uuid", "Chip ID", "ASIC", "Workgroup Max", "Wavefront", "SIMDs", "Shader", "Cacheline")))
        print(t[:6000], flush=True)
    except Exception as e:
        print(cmd[0], "failed:", e)
tot = run(1024, 80, R, tag="config 2")
tot += run(1024, 80, max(2, R // 2), opts=OA.ipopt_opts(), tag="config 2, IPOPT configuration")
for B in (64, 128, 256, 512, 768):
    tot += run(B, 80, max(2, R // 2), tag="smaller batch")
tot += run(2048, 80, 3, tag="two-launch schedule")
tot += run_copies(1024, 80, 3, tag="four copies in flight")
tot += run_copies(1024, 80, 3, opts=OA.ipopt_opts(), tag="four copies in flight, IPOPT configuration")
print("TOTAL differing (instance, run) pairs", tot)
