#!/bin/bash
# round 3, job o: quadcopter forward sweep with the trajectory addressed as LDS (no flat accesses); gather depth 6 / 10
mkdir -p gpurun_out/r3o; O=$PWD/gpurun_out/r3o; R=$PWD; C=$R/obca_amd/csrc
for V in prof prof_d10; do for B in 64 1024; do echo "== $V B $B"; OBCA_HIP_LIBRARY=$C/libobca_hip_$V.so timeout 300 python tools/quad_gpu.py $B 2>&1 | grep -v "^ric:\|^init" | cut -c1-200; done; done > $O/fwd_variants.txt 2>&1; cat $O/fwd_variants.txt
timeout 1500 python -m pytest tests/test_gpu_quad_parity.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest_quad.log; cat $O/pytest_quad.log
timeout 900 python bench.py --config 4 --no-cpu-baseline --no-host-rate --no-pmc --steps 40 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); r=d['roofline']; k=d['config']; print('config 4 value', d['value'], 'ms', d['ms_per_step'], 'validated', k['converged'], '/', k['instances'], 'iters', k['mean_iterations'], 'passes', k['mean_passes'], 'kernel_ms', r['kernel_ms'])" || tail -5 $O/bench_cfg4.err
