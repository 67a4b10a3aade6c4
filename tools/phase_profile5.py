"""Per-phase shader-cycle breakdown of the IPM kernel on the config-5 distribution (1-10 obstacles; needs the -DOBCA_PROFILE build: OBCA_HIP_LIBRARY=.../libobca_hip_prof.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; N = 80
bt = S.make_mixed_batch(B, N, min_obstacles=1)
xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
b.solve(); b.solve()
out = b.download(); pc = b.phase_cycles()
names = "init asm_obs asm_stage ric_bwd border_cl fwd_seq bs_stage bs_obs trial apply other".split()
passes = out['info'][:, 1] + out['info'][:, 6]
nob = np.array([len(v) for v in bt['vOb']]); vmax = np.array([int(np.max(v)) for v in bt['vOb']])
print('kernel ms', b.kernel_ms(), 'B', B, 'converged', int((out['exitflag'] == 1).sum()), 'mean iters', out['iters'].mean(), 'mean passes', passes.mean(), 'max passes', passes.max())
tot = pc[:, :len(names)].sum(1)
print('cycles per pass: mean %.0f' % (tot / passes).mean())
for i, n in enumerate(names):
    print('%-10s %5.1f%%   cycles/pass %8.0f' % (n, 100 * pc[:, i].sum() / tot.sum(), (pc[:, i] / passes).mean()))
for cls, sel in (("rows <= 2", vmax <= 2), ("rows 3-4", vmax > 2)):
    if sel.any():
        print('%s: %d instances, mean obstacles %.1f, cycles per pass %.0f (blocks: assembly %.0f, back-substitution %.0f)' % (cls, sel.sum(), nob[sel].mean(), (tot[sel] / passes[sel]).mean(), (pc[sel, 1] / passes[sel]).mean(), (pc[sel, 7] / passes[sel]).mean()))
