import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
B = 1024; N = 80
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
res = []
for r in range(3):
    b.solve(); o = b.download(); res.append((o['iters'].copy(), o['obj'].copy(), o['xp'].copy(), o['info'].copy()))
print('run-to-run identical:', all(np.array_equal(res[0][0], r[0]) and np.array_equal(res[0][2], r[2]) for r in res[1:]))
np.savez(sys.argv[1], iters=res[0][0], obj=res[0][1], xp=res[0][2], info=res[0][3])
if len(sys.argv) > 2:
    a = np.load(sys.argv[2])
    d = np.flatnonzero(a['iters'] != res[0][0])
    print('vs', sys.argv[2], 'iters differ at', len(d), 'instances', d[:10], a['iters'][d[:10]], res[0][0][d[:10]], 'max dx', np.abs(a['xp'] - res[0][2]).max())
    np.set_printoptions(linewidth=200, precision=4, suppress=False)
    for i in d[:6]:
        print(i, 'A', a['info'][i]); print(i, 'B', res[0][3][i])
