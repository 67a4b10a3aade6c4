import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, obca_amd as OA
from obca_amd import scenarios as S
B, N = 16, 80
bt = S.make_batch(S.BACKWARDS, B, N); xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"])
print(os.environ.get("OBCA_HIP_LIBRARY", "default"), 'exitflag', out['exitflag'], 'iters', out['iters'], 'nreg', out['info'][:, 6])
