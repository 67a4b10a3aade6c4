#!/bin/bash
# round 4, job L: the two powers of the filter line search in one pass through pow() (two lanes): same-box A/B against the previous build, parking and quadcopter; results must be bit-identical
mkdir -p gpurun_out/r4l
O=$PWD/gpurun_out/r4l; C=$PWD/obca_amd/csrc
for rep in 1 2 3; do for L in libobca_hip_prev.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$L pipelined', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'sync', d['config']['single_batch_sync_solves_per_s'], d['config']['converged'])" | tee -a $O/ab_pow_pair.txt
done; done
for L in libobca_hip_prev.so libobca_hip.so; do OBCA_HIP_LIBRARY=$C/$L timeout 200 python - <<'PY' | tee -a $O/ab_pow_pair.txt
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import obca_amd
from obca_amd import scenarios as S
from obca_amd.api import QuadBatch, Context
B, N = 1024, 60
bt = S.make_quad_batch(B, N, random_endpoints=True)
qb = QuadBatch(Context(0), B, N); qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"])
qb.solve(); ms = []
for _ in range(6): qb.solve(); ms.append(qb.kernel_ms())
out = qb.download()
p = S.make_batch(S.BACKWARDS, 1024, 80, seed=20260925); xWS = p["xWS"].copy(); xWS[:, 0, :] = p["x0"]
po = obca_amd.parking_signed_dist_batch(p["x0"], p["xF"], 80, p["Ts"], p["L"], p["ego"], p["XYbounds"], p["vOb"], p["A"], p["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, p["uWS"])
print(os.path.basename(os.environ["OBCA_HIP_LIBRARY"]), "quad kernel_ms %.3f" % np.median(ms), "iters", int(out["iters"].sum()), "checksum %.15e" % float(np.abs(out["xp"]).sum()), "| parking iters", int(po["iters"].sum()), "checksum %.15e" % float(np.abs(np.asarray(po["xp"])).sum()))
PY
done
