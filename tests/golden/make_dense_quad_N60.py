"""Dense Algorithm A (oracle/ipm_ref80.py: attempt) on the reference's UN-reformulated quadcopter NLP (oracle/ipm_ref60_quad.py).  NO fixture of it is committed: unlike the parking
problem (tests/golden/dense_N80.npz) the quadcopter NLP has many local solutions and the dense solve -- another scaling, slacks on every bound -- ends in another one than the
C oracle (`--check`, N = 10, 13 minutes: dense 26.93, oracle 19.54, both "Optimal", and the un-reformulated model accepts the oracle's point with f = 19.54).  The pin of the
quadcopter reformulations is therefore the certificate of tests/test_pin_cpu.py (the model evaluated at the returned solutions); this script stays as the way to repeat the
experiment:  python tests/golden/make_dense_quad_N60.py --check   |   python tests/golden/make_dense_quad_N60.py [count]   (writes tests/golden/dense_quad_N60.npz, ~1 h per instance)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ipm_ref80 as D
import ipm_ref60_quad as DQ
from obca_amd import scenarios as S

OUT = os.path.join(ROOT, "tests", "golden", "dense_quad_N60.npz")


def run(x0, xF, N, Ts, R, ob, xWS, timeWS, verbose=False):
    nlp = DQ.RefQuadNLP(x0, xF, N, Ts, R, ob)
    v0 = nlp.start(xWS, timeWS, DQ.box_duals(xWS, ob))
    o = DQ.QuadOpts(); o.verbose = verbose
    t0 = time.time(); v, st, stats = D.attempt(nlp, v0, o); dt = time.time() - t0
    x, ts, u, lam, s = (a.numpy() for a in nlp.split(__import__("torch").tensor(v)))
    return dict(status=st, xp=x.T.copy(), up=u.T.copy(), ts=ts.copy(), slack=s.T.copy(), stats=stats, seconds=dt)


if "--check" in sys.argv:
    import oracle_quad as Q
    N = 10; Ts = 2.0
    xWS = Q.warm_start(Q.X0, Q.XF, N, [(1.6, 1.4, 0.3), (2.9, 1.9, 0.3), (6.6, 4.5, 2.5), (7.9, 4.5, 2.5)])
    r = run(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, verbose=True)
    oo = Q.default_opts(); oo.max_soc = 4; oo.lsq_init = 1
    c = Q.quadcopter_signed_dist(Q.X0, Q.XF, N, Ts, Q.EGO_R, Q.OB_CLAMPED, xWS, 1.0, opts=oo)
    print(r["status"], r["stats"], "%.1f s" % r["seconds"])
    print("oracle", c["exitflag"], c["iters"], c["obj"], "dense obj", r["stats"]["obj"], "dt", abs(c["t"] - r["ts"][0]), "du", np.abs(c["up"] - r["up"]).max(), "dx", np.abs(c["xp"] - r["xp"]).max())
    sys.exit(0)

count = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
N = 60
q = S.make_quad_batch(max(8, count), N, seed=20260925, random_endpoints=True)
have = dict(np.load(OUT, allow_pickle=False)) if os.path.exists(OUT) else None
rows = [] if have is None else [{k: have[k][i] for k in have if k not in ("N", "Ts", "R", "ob")} for i in range(len(have["idx"]))]
for i in range(count):
    if any(int(r["idx"]) == i for r in rows):
        continue
    r = run(q["x0"][i], q["xF"][i], N, q["Ts"], q["R"], q["ob"], q["xWS"][i], q["timeWS"])
    st = r["stats"]
    print("quad", i, r["status"], {k: st[k] for k in ("iters", "reg", "soc", "soc_acc", "obj", "obj_scaling", "rows_scaled")}, "%.0f s" % r["seconds"], flush=True)
    rows.append(dict(idx=i, x0=q["x0"][i], xF=q["xF"][i], xWS=q["xWS"][i], xp=r["xp"], up=r["up"], ts=r["ts"], slack=r["slack"], obj=st["obj"], iters=st["iters"], reg=st["reg"],
                     soc=st["soc"], soc_acc=st["soc_acc"], exitflag=int(r["status"] == "Optimal"), seconds=r["seconds"], obj_scaling=st["obj_scaling"]))
    np.savez_compressed(OUT, N=N, Ts=q["Ts"], R=q["R"], ob=q["ob"], **{k: np.array([rw[k] for rw in rows]) for k in rows[0]})
