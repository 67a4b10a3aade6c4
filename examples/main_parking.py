"""
main.jl-equivalent driver (AutonomousParking/main.jl:36-330, without the plots) on the MI355X path:
  scenario table -> obstHrep -> Hybrid A* warm start -> ParkingDist (collision-free) -> ParkingSignedDist (min-penetration) -> validate.
Usage:  python examples/main_parking.py [backwards|parallel] [N] [--reference-planner]
  --reference-planner: the warm start as main.jl builds it (BASELINE config 1): the reference's own Hybrid A* restated (REFERENCE mode of the planner library, point-cloud
  obstacles of main.jl:111-133 / 172-198), speed profile, veloSmooth, steering, every third sample; the horizon then follows from the path length (N argument ignored).
Needs libobca_hip.so and a gfx950 device (no CPU fallback); the planner and the validation are host-side numpy / C++.
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import obca_amd
from obca_amd import scenarios as S, planner as PL, validate as V


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]; ref_planner = "--reference-planner" in sys.argv
    name = args[0] if len(args) > 0 else "backwards"
    N = int(args[1]) if len(args) > 1 else 80
    sc = S.BACKWARDS if name == "backwards" else S.PARALLEL
    A, b, vOb = S.scenario_hrep(sc)                                        # main.jl:99-108 / 151-162: obstHrep, vObMPC = vOb - 1
    nOb = len(vOb); x0, xF = sc["x0"], sc["xF"]
    t0 = time.time()
    if ref_planner:
        ws = PL.reference_warm_start(sc, x0, xF)                           # main.jl:216-252 as it stands
        if ws is not None:
            N, ws = ws[0], ws[1:4]
    else:
        ws = PL.warm_start(sc, x0, xF, N)
    t_plan = time.time() - t0
    if ws is None:
        print("planner: no path"); return 1
    Ts, xWS, uWS = ws
    rx, ry, ryaw = xWS[:, 0], xWS[:, 1], xWS[:, 2]
    print("scenario %s: N = %d, Ts = %.3f, Hybrid A* %.2f s" % (name, N, Ts, t_plan))
    for label, fn, dist in (("ParkingDist      ", obca_amd.ParkingDist, True), ("ParkingSignedDist", obca_amd.ParkingSignedDist, False)):
        xp, up, ts, ef, t, lp, npp = fn(x0, xF, N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, nOb, vOb, A, b, rx, ry, ryaw, 0, xWS, uWS)   # main.jl:258,269
        ok, viol = V.validate_parking(x0, xF, N, Ts, S.L_WHEELBASE, S.EGO, S.XYBOUNDS, vOb, A, b, xp, up, ts, lp, npp, dist=dist,
                                      tol=5e-5 if dist else 1e30)
        print("%s exitflag %d  %.1f ms  timeScale %.3f  penetration %+.4f m  %s" % (label, ef, 1e3 * t, float(np.ravel(ts)[0]), viol["penetration"],
              ("validate: ok" if ok else "validate: FAILED %s" % {k: v for k, v in viol.items() if v > 5e-5}) if dist else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
