"""Writes tests/golden/reference_instances.json: the inputs julia/bench_reference.jl feeds to the UNTOUCHED reference (ParkingSignedDist.jl + IPOPT) on a box that
has Julia + Ipopt -- the first 8 instances of BASELINE config 2 (backwards parking, N=80, seed 20260925) and the 4 config-3 instances of oracle_cfg3.npz (parallel
parking, Hybrid A* warm starts).  Run from the repo root: python tests/golden/make_reference_instances.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from obca_amd import scenarios as S
OUT = os.path.dirname(os.path.abspath(__file__))
N = 80
sets = []
bt = S.make_batch(S.BACKWARDS, 8, N)
A, b, v = S.scenario_hrep(S.BACKWARDS)
inst = []
for i in range(8):
    xWS = bt["xWS"][i].copy(); xWS[0] = bt["x0"][i]
    inst.append(dict(x0=bt["x0"][i].tolist(), xF=bt["xF"][i].tolist(), Ts=float(bt["Ts"][i]), xWS=xWS.tolist(), uWS=bt["uWS"][i].tolist()))
sets.append(dict(scenario="backwards", L=S.L_WHEELBASE, ego=S.EGO.tolist(), XYbounds=S.XYBOUNDS.tolist(), nOb=len(v), vOb=[int(x) for x in v], A=A.tolist(), b=b.tolist(), N=N, instances=inst))
g = np.load(os.path.join(OUT, "oracle_cfg3.npz"))
A, b, v = S.scenario_hrep(S.PARALLEL)
inst = [dict(x0=g["x0"][i].tolist(), xF=g["xF"][i].tolist(), Ts=float(g["Ts"][i]), xWS=g["xWS"][i].tolist(), uWS=g["uWS"][i].tolist()) for i in range(4)]
sets.append(dict(scenario="parallel", L=S.L_WHEELBASE, ego=S.EGO.tolist(), XYbounds=S.XYBOUNDS.tolist(), nOb=len(v), vOb=[int(x) for x in v], A=A.tolist(), b=b.tolist(), N=N, instances=inst))
for s_ in sets:
    json.dump(s_, open(os.path.join(OUT, f"reference_instances_{s_['scenario']}.json"), "w"))
    print(s_["scenario"], len(s_["instances"]))
