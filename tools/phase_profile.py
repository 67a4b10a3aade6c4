"""Per-phase shader-cycle breakdown of the IPM kernel (needs the -DOBCA_PROFILE build: OBCA_HIP_LIBRARY=.../libobca_hip_prof.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; N = 80
opts = OA.ipopt_opts() if len(sys.argv) > 2 and sys.argv[2].startswith("ipopt") else None      # second argument "ipopt": the reference's IPOPT configuration ("ipopt-norestore": without its block restoration)
if opts is not None and sys.argv[2] == "ipopt-norestore": opts.restoration = 0
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
ctx = OA.Context(0); b = OA.Batch(ctx, B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
b.solve(opts=opts); b.solve(opts=opts)
out = b.download(); pc = b.phase_cycles()
names = "init asm_obs asm_stage ric_bwd border_cl fwd_seq bs_stage bs_obs trial apply other ric_p1 ric_p2".split()
passes = out['info'][:, 1] + out['info'][:, 6]
print('kernel ms', b.kernel_ms(), 'B', B, 'mean iters', out['iters'].mean(), 'mean passes', passes.mean(), 'max passes', passes.max())
tot = pc[:, :11].sum(1)      # (slots 11, 12: per-stage counters of the FINE build, else the residency stamps tools/load_profile.py reads)
slow = int(np.argmax(tot)); print('slowest instance %d: iterations %d, inertia rungs %d, total cycles %.0f; per phase:' % (slow, out['info'][slow, 1], out['info'][slow, 6], tot[slow]), {n: int(pc[slow, i]) for i, n in enumerate(names)}, 'soc/rebuild/recalc', pc[slow, 13:16].tolist())
print('cycles per pass: mean %.0f  (slowest instance %.0f total cycles = %.1f ms @2.4GHz)' % ((tot / passes).mean(), tot.max(), tot.max() / 2.4e6))
print('options:', 'reference IPOPT configuration' if opts is not None else 'throughput defaults', '| per solve: second-order corrections tried %.2f, Newton systems rebuilt %.2f, multiplier re-estimates %.2f' % (pc[:, 13].mean(), np.floor(pc[:, 14]).mean(), pc[:, 15].mean()))
for i, n in enumerate(names):
    print('%-10s %5.1f%%   cycles/pass %8.0f' % (n, 100 * pc[:, i].sum() / tot.sum(), (pc[:, i] / passes).mean()))
