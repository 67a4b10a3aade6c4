import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, obca_amd as OA
from obca_amd import scenarios as S
B, N = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 80
bt = S.make_batch(S.BACKWARDS, B, N); xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
b.solve(); out = b.download(); pc = b.phase_cycles()
print('exitflag', out['exitflag'], 'iters', out['iters'])
print('checksum rel diff', pc[:, 13]); print('Bm max diff', pc[:, 14]); print('ok flags (100 + 10 lds + mfma)', pc[:, 15])
