"""what each rank of an 8-GPU weak-scaling run would see: kernel time and the longest solve of its own batch (seed 20260925 + rank)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
B = 1024; N = 80
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
for r in range(8):
    bt = S.make_batch(S.BACKWARDS, B, N, seed=20260925 + r)
    xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
    b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
    ms = []
    for _ in range(3):
        b.solve(); ms.append(sum(b.kernel_ms()))
    out = b.download(); p = out['info'][:, 1] + out['info'][:, 6]
    print("rank %d: ipm+dualws %.2f ms, converged %d, passes mean %.1f p99 %.0f max %d (instance %d)" % (r, min(ms), (out['exitflag'] == 1).sum(), p.mean(), np.percentile(p, 99), p.max(), p.argmax()))
