"""distribution of factorisation passes over the config-2 batch: which instances make the tail that sets the kernel time"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; N = 80
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
b.solve(); out = b.download()
info = out['info']; passes = info[:, 1] + info[:, 6]
print('kernel ms', b.kernel_ms())
print('passes percentiles 50/90/99/max', np.percentile(passes, [50, 90, 99, 100]))
print('hist', np.histogram(passes, bins=[0, 20, 30, 40, 50, 60, 80, 100, 150, 500])[0])
idx = np.argsort(-passes)[:16]
for i in idx:
    print(i, 'iters', int(info[i, 1]), 'nreg', int(info[i, 6]), 'status', int(info[i, 0]), 'ef', int(info[i, 7]), 'x0', np.round(bt['x0'][i], 3))
