#!/bin/bash
# round 5, job E: NaN-poisoned work buffers / LDS at full occupancy and beyond (1 024 resident, 4 096 through the two-launch schedule); per-phase clocks under both option sets
mkdir -p gpurun_out/r5e
O=$PWD/gpurun_out/r5e; C=$PWD/obca_amd/csrc
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python tools/determinism.py 6 > $O/poison_1024.txt 2>&1; tail -n 2 $O/poison_1024.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python - > $O/poison_4096.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
N, B = 80, 4096
bt = S.make_mixed_batch(B, N, seed=20260925, min_obstacles=1)
xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
for opts, name in ((None, "default"), (OA.ipopt_opts(), "ipopt")):
    ref = None
    for r in range(3):
        out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], N, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], opts=opts)
        bad = int((~np.isfinite(out["info"])).any(axis=1).sum())
        if ref is None: ref = out
        same = np.array_equal(out["info"], ref["info"])
        print("poisoned build, config 5 batch of 4096,", name, "options, run", r, ": solved", int((out["exitflag"] == 1).sum()), "non-finite info rows", bad, "same bits as run 0:", same)
PY
tail -n 6 $O/poison_4096.txt
for o in default ipopt; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/phase_profile.py 1024 $o > $O/phase_cycles_B1024_$o.txt 2>&1; cat $O/phase_cycles_B1024_$o.txt; done
