"""GPU vs oracle on the full bench batches of BASELINE configs 3 and 5 (rank 0's batch of `bench.py --config 3 / 5`): how many exit flags / iteration counts / trajectories agree.
The numbers decide what tests/test_gpu_parity.py asserts at size.  python tools/parity_census.py 3 5
With --ipopt-options the oracle additionally runs with the two IPOPT semantics the HIP kernels do not have (second-order correction max_soc = 4, recalc_y = yes: oracle options,
DESIGN.md section 2) and the GPU results are compared with THAT run: do the solved sets and the optima change?
With --gpu-ipopt-options the kernels run with their own IPOPT switches (max_soc = 4, recalc_y, lsq_init) and are compared with the oracle running the same options: parity at size."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "oracle"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np
SEED = 20260925


def main():
    from obca_amd import scenarios as S
    cfgs = [int(a) for a in sys.argv[1:] if not a.startswith("-")] or [3, 5]
    batches = {}
    for c in cfgs:      # (planned before HIP is up: fork-safe)
        t0 = time.time()
        batches[c] = S.make_batch(S.PARALLEL, 2048, 80, seed=SEED, goal_jitter=True) if c == 3 else (S.make_mixed_batch(4096, 80, seed=SEED, min_obstacles=1) if c == 5 else S.make_batch(S.BACKWARDS, 1024, 80, seed=SEED))
        print("config", c, "batch made in %.1f s" % (time.time() - t0), flush=True)
    import obca_amd as OA
    import oracle_pool
    for c in cfgs:
        bt = batches[c]; N = 80; B = len(bt["x0"])
        xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
        t0 = time.time()
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
        t1 = time.time()
        ref = oracle_pool.parking_oracle_all(bt, xWS) if c != 5 else oracle_pool.mixed_oracle_all(bt, xWS)
        t2 = time.time()
        ef_bad = it_bad = x_bad = 0; worst = 0.0; wf = 0.0; its = []
        for r in ref:
            i, ef, it, obj, xp = r[0], r[1], r[2], r[3], r[4]
            its.append(it)
            if out["exitflag"][i] != ef: ef_bad += 1; continue
            if out["iters"][i] != it:
                it_bad += 1; print("  iteration mismatch: instance", i, "gpu", out["iters"][i], "oracle", it, "obj", out["obj"][i], obj, "dx", np.abs(out["xp"][i] - xp).max()); continue
            dx = np.abs(out["xp"][i] - xp).max(); worst = max(worst, dx); wf = max(wf, abs(out["obj"][i] - obj) / max(1, abs(obj)))
            if dx > 1e-6: x_bad += 1
        if "--ipopt-options" in sys.argv:
            ref2 = oracle_pool.parking_oracle_all(bt, xWS, switches=(4, 1, 0)) if c != 5 else oracle_pool.mixed_oracle_all(bt, xWS, switches=(4, 1, 0))
            efd = itd = 0; wx = wf2 = 0.0; ndiff = 0
            for r in ref2:
                i, ef, it, obj, xp = r[0], r[1], r[2], r[3], r[4]
                efd += int(out["exitflag"][i] != ef); itd += int(out["iters"][i] != it)
                if ef == 1 and out["exitflag"][i] == 1:
                    dxi = np.abs(out["xp"][i] - xp).max(); wx = max(wx, dxi); wf2 = max(wf2, abs(out["obj"][i] - obj) / max(1, abs(obj))); ndiff += int(dxi > 1e-3)
            print("config %d vs oracle WITH max_soc=4 + recalc_y: exit-flag differences %d  iteration-count differences %d  instances ending in another local solution (|dx| > 1e-3) %d  worst |dx| %.2e  worst rel. objective difference %.2e" % (c, efd, itd, ndiff, wx, wf2), flush=True)
        if "--gpu-ipopt-options" in sys.argv:      # the kernels WITH their IPOPT switches against the oracle with the same ones: parity at size (round 3: the switches exist on both sides)
            o = OA.ipopt_opts()
            out3 = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], opts=o)
            ref3 = oracle_pool.parking_oracle_all(bt, xWS, switches=oracle_pool.IPOPT) if c != 5 else oracle_pool.mixed_oracle_all(bt, xWS, switches=oracle_pool.IPOPT)
            efd = itd = nx = 0; wx = 0.0
            for r in ref3:
                i, ef, it, obj, xp = r[0], r[1], r[2], r[3], r[4]
                efd += int(out3["exitflag"][i] != ef); itd += int(out3["iters"][i] != it)
                if ef == 1 and out3["exitflag"][i] == 1 and out3["iters"][i] == it:
                    dxi = np.abs(out3["xp"][i] - xp).max(); wx = max(wx, dxi); nx += int(dxi > 1e-6)
            print("config %d, max_soc = 4 + recalc_y + lsq_init on BOTH sides: exit-flag mismatches %d  iteration mismatches %d  |dx| > 1e-6 %d  worst |dx| %.2e  (solved: gpu %d oracle %d)"
                  % (c, efd, itd, nx, wx, int((out3["exitflag"] == 1).sum()), sum(1 for r in ref3 if r[1] == 1)), flush=True)
        print("config %d: B %d  gpu %.2f s  oracle %.1f s  exitflag==1 gpu %d oracle %d  exit-flag mismatches %d  iteration mismatches %d  |dx|>1e-6 %d  worst dx %.2e  worst df %.2e  max iters %d"
              % (c, B, t1 - t0, t2 - t1, int((out["exitflag"] == 1).sum()), sum(1 for r in ref if r[1] == 1), ef_bad, it_bad, x_bad, worst, wf, max(its)), flush=True)


if __name__ == "__main__":
    main()
