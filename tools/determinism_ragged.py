"""Bit-reproducibility of the HIP path on RAGGED batches under concurrency (the driver's round-4 GPU run failed exactly here:
tests/test_gpu_multi.py::test_chunked_host_call_equals_resident_batch[1000-2]).

One device-resident solve of a mixed batch (1-16 / 3-10 obstacles per instance) is the reference; then R host-pointer calls cut into chunks over 1-4 concurrent worker
lanes (streams), fresh contexts and reused ones, shuffled chunk sizes -- every result must equal the reference bit for bit.  Reports WHICH fields of WHICH instances differ.

  python tools/determinism_ragged.py [R] [N] [B] [opts: default|reference]      (OBCA_HIP_LIBRARY selects a diagnostic build)
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd as OA
from obca_amd import scenarios as S

R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 150
OPT = sys.argv[4] if len(sys.argv) > 4 else "default"
opts = OA.ipopt_opts() if OPT == "reference" else None
INFO = ("status", "iters", "obj", "pinf", "dinf", "mu", "nreg", "exitflag")


def resident(bt, xWS):
    ctx = OA.Context(0)
    b = OA.Batch(ctx, len(bt["x0"]), N)
    b.upload(bt["x0"], bt["xF"], bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
    outs = []
    for _ in range(3):
        b.solve(opts=opts) if opts is not None else b.solve()
        outs.append(b.download())
    b.close(); ctx.close()
    return outs


def diff(a, b, tag):
    bad = []
    if not np.array_equal(a["info"], b["info"]):
        w = np.argwhere(a["info"] != b["info"])
        for i in sorted(set(w[:, 0]))[:4]:
            cols = [INFO[c] for c in w[w[:, 0] == i, 1]]
            bad.append("inst %d info%s: %s | %s" % (i, cols, a["info"][i].tolist(), b["info"][i].tolist()))
    for k in ("xp", "up", "timeScale"):
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
            d = np.abs(np.asarray(a[k]) - np.asarray(b[k])).reshape(len(a["info"]), -1).max(axis=1)
            bad.append("%s differs in %d instances (first %d, max %.3e)" % (k, int((d > 0).sum()), int(np.flatnonzero(d > 0)[0]) if (d > 0).any() else -1, d.max()))
    for k in ("lp", "np", "sl"):
        nb = sum(0 if np.array_equal(a[k][i], b[k][i]) else 1 for i in range(len(a["info"])))
        if nb:
            bad.append("%s differs in %d instances" % (k, nb))
    if bad:
        print("  MISMATCH", tag, "::", " ;; ".join(bad), flush=True)
    return bool(bad)


def main():
    rng = np.random.default_rng(11)
    total_bad = 0
    for seed, gen in ((3, "mixed"), (5, "mixed")):
        bt = S.make_mixed_batch(B, N, seed=seed)
        xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
        refs = resident(bt, xWS)
        ref = refs[0]
        nb = sum(diff(r, ref, "resident repeat %d seed %d" % (i, seed)) for i, r in enumerate(refs[1:]))
        combos = [(1000, 2), (37, 3), (20, 4), (64, 1), (1000, 2), (11, 4), (75, 2), (50, 3)]
        args = (bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
        kw = dict(opts=opts) if opts is not None else {}
        for r in range(R):
            chunk, slots = combos[r % len(combos)]
            if r >= len(combos) and r % 3 == 0:
                chunk = int(rng.integers(5, 160)); slots = int(rng.integers(1, 5))
            os.environ["OBCA_CHUNK"] = str(chunk); os.environ["OBCA_SLOTS"] = str(slots)
            ctx = OA.Context(0)
            for rep in range(2):                              # second call: cached lane batches, another chunk -> lane assignment
                out = OA.parking_signed_dist_batch(*args, device=ctx, **kw)
                nb += diff(out, ref, "seed %d run %d chunk %d slots %d call %d" % (seed, r, chunk, slots, rep))
            ctx.close()
        del os.environ["OBCA_CHUNK"]; del os.environ["OBCA_SLOTS"]
        print("seed", seed, ": converged", int((ref["exitflag"] == 1).sum()), "of", B, "; mismatching calls", nb, "of", 2 * R + 2, flush=True)
        total_bad += nb
    print(os.environ.get("OBCA_HIP_LIBRARY", "default library"), "opts", OPT, "N", N, "B", B, "TOTAL mismatching calls", total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
