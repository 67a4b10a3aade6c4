#!/bin/bash
# round 2, job h: A/B of the fp32 Riccati factorisation (-DOBCA_RICCATI_FP32) against the fp64 default: config 2 and config 5 bench lines, per-phase clocks
mkdir -p gpurun_out/r2h
O=$PWD/gpurun_out/r2h; R=$PWD
for V in default fp32; do
  L=$R/obca_amd/csrc/libobca_hip_$V.so; [ "$V" = "default" ] && L=$R/obca_amd/csrc/libobca_hip.so
  OBCA_HIP_LIBRARY=$L timeout 300 python bench.py --no-cpu-baseline --steps 80 > $O/cfg2_$V.json 2>/dev/null
  OBCA_HIP_LIBRARY=$L timeout 600 python bench.py --no-cpu-baseline --config 5 --steps 16 > $O/cfg5_$V.json 2>/dev/null
  OBCA_HIP_LIBRARY=$L timeout 600 python bench.py --no-cpu-baseline --config 3 --steps 24 > $O/cfg3_$V.json 2>/dev/null
done
python - <<'PY'
import json
for c in (2,3,5):
    for v in ("default","fp32"):
        d=json.load(open(f"gpurun_out/r2h/cfg{c}_{v}.json")); k=d["config"]; r=d["roofline"]
        print("config",c,v,"value",d["value"],"ms/step",d["ms_per_step"],"kernel_ms",r["kernel_ms"],"validated",k["converged"],"/",k["instances"],"exitflag1",k["exitflag_ok"],"mean iters",k["mean_iterations"],"mean passes",k["mean_passes"],"max iters",k["max_iterations"])
PY
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_fp32prof.so timeout 300 python tools/phase_profile.py 64 | tee $O/phase_fp32_B64.txt
