#!/bin/bash
# round 5, job Z: the poisoned build with an 8 KB NaN guard behind the dynamic LDS block: does anything read beyond the block?  (alone; product library's results as reference)
mkdir -p gpurun_out/r5z
python - 2>&1 <<'PY' | tee gpurun_out/r5z/guard.txt | cut -c1-220
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
N, B = 80, 1024
bt = S.make_batch(S.BACKWARDS, B, N)
xWS = bt['xWS'].copy(); xWS[:, 0, :] = bt['x0']
b = OA.Batch(OA.Context(0), B, N)
b.upload(bt['x0'], bt['xF'], bt['Ts'], bt['L'], bt['ego'], bt['XYbounds'], bt['vOb'], bt['A'], bt['b'], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt['uWS'])
for name, o in (('def', None), ('ref', OA.ipopt_opts())):
    b.solve(opts=o); r = b.download()
    np.savez('/tmp/guard_%s_%s.npz' % (sys.argv[1], name), info=r['info'], xp=r['xp'])
    print(sys.argv[1], name, 'solved', int((r['exitflag'] == 1).sum()), 'non-finite info rows', int((~np.isfinite(r['info'])).any(axis=1).sum()), flush=True)
"""
C = os.path.join(os.getcwd(), "obca_amd", "csrc")
for tag, lib in (("product", "libobca_hip.so"), ("p15", "variants/libobca_hip_poison_p15.so"), ("p3", "variants/libobca_hip_poison_p3.so"), ("p6", "variants/libobca_hip_poison_p6.so"), ("p5", "variants/libobca_hip_poison_p5.so"), ("old", "variants/libobca_hip_poison.so")):
    subprocess.run([sys.executable, "-c", code, tag], env=dict(os.environ, OBCA_HIP_LIBRARY=os.path.join(C, lib)))
import numpy as np
for tag in ("p15", "p3", "p6", "p5", "old"):
    for name in ("def", "ref"):
        a = np.load("/tmp/guard_product_%s.npz" % name); p = np.load("/tmp/guard_%s_%s.npz" % (tag, name))
        d = int(((a["info"] != p["info"]).any(axis=1) | (np.abs(a["xp"] - p["xp"]).reshape(1024, -1).max(axis=1) > 0)).sum())
        print("poison parts %s (0 none: only the 8 KB larger LDS allocation; 1 HBM buffers, 2 static LDS, 4 dynamic LDS, 8 guard), options %s: instances that differ from the product build: %d" % (tag, name, d))
PY
