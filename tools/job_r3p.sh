#!/bin/bash
# round 3, job p: quadcopter kernel with the line search fused into the assembly
mkdir -p gpurun_out/r3p; O=$PWD/gpurun_out/r3p; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests/test_gpu_quad_parity.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest_quad.log; tail -8 $O/pytest_quad.log
timeout 900 python bench.py --config 4 --no-cpu-baseline --no-host-rate --steps 40 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); r=d['roofline']; k=d['config']; print('config 4 value', d['value'], 'ms', d['ms_per_step'], 'validated', k['converged'], '/', k['instances'], 'iters', k['mean_iterations'], 'passes', k['mean_passes'], 'kernel_ms', r['kernel_ms'], 'traffic', r['traffic'])" || tail -5 $O/bench_cfg4.err
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/quad_gpu.py $B 2>&1 | grep -v "^ric:\|^init" | cut -c1-200 > $O/quad_phase_B$B.txt; cat $O/quad_phase_B$B.txt; done
