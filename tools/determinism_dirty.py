"""Does the solver read anything it has not written?  OBCA_DIRTY=mask makes the library launch, in front of every interior-point launch, a kernel that leaves a bit pattern
(OBCA_DIRTY_VALUE, a double; default NaN) in the vector registers (1), the accumulation registers (2), the CU's LDS (4) and scratch memory (8) -- what a following workgroup inherits
from whoever used the SIMD before it.  The bench batch is solved `R` times per (mask, value) and compared bit for bit with a solve without the dirtying kernel.
Round 5 ran this with NaN only (job R: 0 differences) and so missed the read that mattered: it went through fmax(), which drops NaN.  1e30 finds it on the unfixed source
(DESIGN.md section 11); the values below are the ones tests/test_gpu_history.py uses.

  python tools/determinism_dirty.py [mask ...]"""
import os, sys, subprocess
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(HERE, ".."))
    import obca_amd as OA
    from obca_amd import scenarios as S
    N, B, R = 80, 1024, int(sys.argv[2])
    bt = S.make_batch(S.BACKWARDS, B, N)
    xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
    b = OA.Batch(OA.Context(0), B, N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    ref = np.load(sys.argv[3])
    for name, opts in (("throughput options", None), ("reference options", OA.ipopt_opts())):
        bad = nan = 0
        for r in range(R):
            b.solve(opts=opts); o = b.download()
            k = "info_ref" if opts is not None else "info_def"; kx = "xp_ref" if opts is not None else "xp_def"
            bad += int(((o["info"] != ref[k]).any(axis=1) | (np.abs(o["xp"] - ref[kx]).reshape(B, -1).max(axis=1) > 0)).sum()); nan += int((~np.isfinite(o["info"])).any(axis=1).sum())
        print("OBCA_DIRTY=%s value %s, %s: %d solves of %d instances, %d (instance, solve) results differ from the clean solve, %d rows with non-finite info" % (os.environ.get("OBCA_DIRTY", "0"), os.environ.get("OBCA_DIRTY_VALUE", "nan"), name, R, B, bad, nan), flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.join(HERE, ".."))
import obca_amd as OA
from obca_amd import scenarios as S
N, B = 80, 1024
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
b = OA.Batch(OA.Context(0), B, N)
b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
b.solve(); d = b.download(); b.solve(opts=OA.ipopt_opts()); r = b.download()
np.savez("/tmp/obca_dirty_ref.npz", info_def=d["info"], xp_def=d["xp"], info_ref=r["info"], xp_ref=r["xp"])
for mask in (sys.argv[1:] or ["0", "15", "1", "2", "4", "8"]):
    for value in ("1e30", "-1e30", "0.5", "nan"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", "3", "/tmp/obca_dirty_ref.npz"], env=dict(os.environ, OBCA_DIRTY=mask, OBCA_DIRTY_VALUE=value))
        if mask == "0":
            break
