"""Two-launch schedule: how well does the first slice predict the stragglers, and what does the schedule gain?  (GPU)"""
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


def run(passes, only=0, reps=3):
    os.environ["OBCA_SLICE_PASSES"] = str(passes); os.environ["OBCA_SLICE_ONLY"] = str(only)
    ms = []
    for _ in range(reps):
        b.solve(); ms.append(b.kernel_ms()[0])
    return min(ms), b.download()


ms0, full = run(0)
passes = full['info'][:, 1] + full['info'][:, 6]
print("single launch: %.2f ms; passes mean %.1f max %d" % (ms0, passes.mean(), passes.max()))
top = np.argsort(-passes)[:16]
for Q in (3, 4, 6, 8, 12):
    msq, part = run(Q, only=1, reps=1)
    info = part['info']; nreg = info[:, 6]; pinf = info[:, 3]; susp = info[:, 0] == 3
    e = np.clip(np.floor(np.log10(np.maximum(pinf, 1e-30))) + 7, 0, 7)
    cls = 8 * np.minimum(nreg, 7) + e
    order = np.argsort(-cls, kind="stable"); rank = np.empty(B, int); rank[order] = np.arange(B)
    ms2, out2 = run(Q)
    same = np.array_equal(out2['info'], full['info']) and np.array_equal(out2['xp'], full['xp'])
    print("Q=%2d slice %.2f ms, parked %4d; ranks of the 16 slowest: %s; two launches %.2f ms (bit-identical to single launch: %s)" %
          (Q, msq, susp.sum(), sorted(rank[top].tolist()), ms2, same))
    print("      corr(passes, nreg_slice) %.2f   remaining-passes of the first 32 dispatched: %s" %
          (np.corrcoef(passes, nreg)[0, 1], (passes[order[:32]] - Q).astype(int).tolist()))
