"""Per-phase shader clocks of the interior-point kernel with the machine kept full (the regime `value` is quoted in: --streams batches in flight) beside the lone launch.
Needs the -DOBCA_PROFILE build: OBCA_HIP_LIBRARY=.../libobca_hip_prof.so python tools/load_profile.py [streams] [steps].
Prints, per phase, clocks per pass alone / under load, and the slot occupancy the wall clock implies: (sum of the instances' clocks of the timed steps) / (resident slots x wall time)
= utilisation x shader clock."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S

def main():
    nS = int(sys.argv[1]) if len(sys.argv) > 1 else 4; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    B, N = 1024, 80; opts = OA.ipopt_opts()
    bt = S.make_batch(S.BACKWARDS, B, N); xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
    bs = []
    for _ in range(nS):
        b = OA.Batch(OA.Context(0), B, N)      # one context = one HIP stream
        b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
        bs.append(b)
    names = "init asm_obs asm_stage ric_bwd border_cl fwd_seq bs_stage bs_obs trial apply other".split()
    bs[0].solve(opts=opts); bs[0].solve(opts=opts); out = bs[0].download(); passes = (out['info'][:, 1] + out['info'][:, 6]).astype(float)
    lone = bs[0].phase_cycles()[:, :len(names)].copy(); lone_ms = bs[0].kernel_ms()[0]
    for k in range(2 * nS): bs[k % nS].solve(opts=opts, sync=False)
    for b in bs: b.sync()
    t0 = time.perf_counter()
    for k in range(steps): bs[k % nS].solve(opts=opts, sync=False)
    for b in bs: b.sync()
    dt = time.perf_counter() - t0
    raw = [b.phase_cycles() for b in bs]
    load = np.mean([r[:, :len(names)] for r in raw], axis=0)      # the last launch of every stream: all of them ran with the machine full
    print("streams %d, %d steps of %d instances: %.3f ms per step = %.1f k solves/s; lone launch %.2f ms" % (nS, steps, B, 1e3 * dt / steps, steps * B / dt / 1e3, lone_ms))
    print("%-10s %12s %12s %7s" % ("phase", "alone", "under load", "ratio"))
    for i, n in enumerate(names):
        a, l_ = (lone[:, i] / passes).mean(), (load[:, i] / passes).mean()
        print("%-10s %12.0f %12.0f %7.2f" % (n, a, l_, l_ / max(a, 1)))
    ta, tl = lone.sum(1), load.sum(1)
    print("%-10s %12.0f %12.0f %7.2f   (clocks per pass, mean over instances)" % ("all", (ta / passes).mean(), (tl / passes).mean(), (tl / passes).mean() / (ta / passes).mean()))
    print("sum of clocks of one batch under load: %.3e; per step and resident slot (1 024): %.3e clocks in %.3f ms -> utilisation x shader clock = %.2f GHz" %
          (tl.sum(), tl.sum() / 1024, 1e3 * dt / steps, tl.sum() / 1024 / (dt / steps) / 1e9))

    # residency on the constant 100 MHz clock (slots 11, 12 of the profiling build): a step launches 1 024 workgroups; the slot time they hold / (1 024 slots x step time) = utilisation
    res = np.array([np.floor(r[:, 12]) / 1e8 for r in raw]); cyc = np.array([r[:, :len(names)].sum(1) for r in raw])
    print("resident time per workgroup: mean %.3f ms (sum %.3f s per launch); shader clock while resident: %.3f GHz (min %.3f max %.3f over instances)" %
          (1e3 * res.mean(), res.sum(1).mean(), (cyc / res).mean() / 1e9, (cyc / res).min() / 1e9, (cyc / res).max() / 1e9))
    print("utilisation of the 1 024 resident slots: %.3f" % (res.sum(1).mean() / (1024 * dt / steps)))
    for r in raw:
        st = (r[:, 11] - r[:, 11].min()) / 1e5; en = (r[:, 11] + np.floor(r[:, 12]) - r[:, 11].min()) / 1e5
        print("  last launch of a stream: workgroup starts spread over %.2f ms (median %.2f), last end %.2f ms; ends: median %.2f, 90 %% %.2f, 99 %% %.2f" %
              (st.max(), np.median(st), en.max(), np.median(en), np.percentile(en, 90), np.percentile(en, 99)))

if __name__ == "__main__":
    main()
