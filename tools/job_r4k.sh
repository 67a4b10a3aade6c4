#!/bin/bash
# round 4, job K: quadcopter kernel, step application with more loads in flight per lane (QAP_R = 6 / 9 / 14 items per chunk): same-box A/B, bit-identical results expected
mkdir -p gpurun_out/r4k
O=$PWD/gpurun_out/r4k; C=$PWD/obca_amd/csrc
for rep in 1 2; do for L in libobca_hip.so libobca_hip_qap9.so libobca_hip_qap14.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 200 python - <<'PY' | tee -a $O/ab_apply_chunk.txt
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
print(os.path.basename(os.environ["OBCA_HIP_LIBRARY"]), "kernel_ms %.3f" % np.median(ms), "iters", int(out["iters"].sum()), "checksum %.12e" % float(np.abs(out["xp"]).sum()))
PY
done; done
