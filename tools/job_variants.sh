#!/bin/bash
# A/B of kernel build variants: libobca_hip_<name>.so for every name given; bench line (pipelined + synchronous kernel time) each
mkdir -p gpurun_out/variants
R=$PWD
for V in "$@"; do
  L=$R/obca_amd/csrc/libobca_hip_$V.so; [ "$V" = "default" ] && L=$R/obca_amd/csrc/libobca_hip.so
  OBCA_HIP_LIBRARY=$L timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$V value',d['value'],'ms',d['ms_per_step'],'kernel_ms',r['kernel_ms'],'conv',d['config']['converged'],'iters',d['config']['mean_iterations'])"
done | tee gpurun_out/variants/$(date +%H%M%S).txt
