"""Who holds the 1 024 wavefront slots when K launches of the config-2 batch are queued at once on K streams (profiling build: OBCA_HIP_LIBRARY=.../libobca_hip_prof.so; set
GPU_MAX_HW_QUEUES >= K): every workgroup's start, resident time and SIMD (slots 11, 12 of the phase counters).  Printed: per-XCD work, idle gaps between consecutive workgroups of a
slot while launches still had workgroups to dispatch, and the utilisation of that window.   python tools/slot_timeline.py [K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S

def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    B, N = 1024, 80; opts = OA.ipopt_opts()
    bt = S.make_batch(S.BACKWARDS, B, N); xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
    bs = []
    for _ in range(K):
        b = OA.Batch(OA.Context(0), B, N)
        b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
        bs.append(b)
    for b in bs: b.solve(opts=opts, sync=False)      # warm-up (first launches allocate scratch)
    for b in bs: b.sync()
    for b in bs: b.solve(opts=opts, sync=False)
    for b in bs: b.sync()
    raw = np.array([b.phase_cycles() for b in bs])                      # K x B x 16
    st = raw[:, :, 11]; dur = np.floor(raw[:, :, 12]); hw = np.rint((raw[:, :, 12] - dur) * 65536).astype(int); pipe = (hw >> 2) & 3; hw &= ~0xc      # HW_ID[7:6] is the pipe the launch's queue sits on, not a place
    t0 = st.min(); st = (st - t0) / 1e5; en = st + dur / 1e5           # ms
    xcc = hw >> 12
    print("%d launches x %d workgroups; distinct slots seen: %d; first start 0, last start %.2f ms, last end %.2f ms" % (K, B, len(np.unique(hw)), st.max(), en.max()))
    print("launch -> pipe:", [np.unique(pipe[k]).tolist() for k in range(K)])
    print("workgroups per XCD:", np.bincount(xcc.ravel(), minlength=8).tolist())
    print("resident ms per XCD:", np.round(np.bincount(xcc.ravel(), weights=(en - st).ravel(), minlength=8), 1).tolist())
    same = [(np.bincount(xcc[k], minlength=8)).tolist() for k in range(min(K, 3))]
    print("workgroups per XCD of the first launches:", same, " blockIdx %% 8 == XCD for %.3f of the workgroups" % np.mean(xcc == (np.arange(B)[None, :] % 8)))
    T = np.percentile(st, 90)           # until here launches still had workgroups to hand out
    gaps = []; busy = 0.0; nslot = 0
    for h in np.unique(hw):
        m = hw == h; s_, e_ = st[m], en[m]; o = np.argsort(s_); s_, e_ = s_[o], e_[o]
        nslot += 1; busy += np.clip(np.minimum(e_, T) - np.minimum(s_, T), 0, None).sum()
        g = s_[1:] - e_[:-1]; gaps.extend(g[s_[1:] <= T].tolist())
    gaps = np.array(gaps)
    print("window 0 .. %.2f ms (90 %% of the workgroups started): utilisation of %d slots %.3f" % (T, nslot, busy / (nslot * T)))
    print("gap between a workgroup's end and the next start on its slot: median %.1f us, mean %.1f us, 90 %% %.1f us, 99 %% %.1f us, max %.1f us; negative (overlap) %d of %d; sum of gaps / slot time %.3f" %
          (1e3 * np.median(gaps), 1e3 * gaps.mean(), 1e3 * np.percentile(gaps, 90), 1e3 * np.percentile(gaps, 99), 1e3 * gaps.max(), (gaps < 0).sum(), len(gaps), gaps.clip(0).sum() / (nslot * T)))
    for x in range(8):
        m = xcc == x
        print("  XCD %d: last start %.2f ms, last end %.2f ms, resident %.1f ms" % (x, st[m].max(), en[m].max(), (en - st)[m].sum()))

if __name__ == "__main__":
    main()
