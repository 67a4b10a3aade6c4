#!/bin/bash
# round 3, job n: quadcopter kernel -- forward sweep on quad-lane sums with a six-stage gather pipeline, item-parallel costate increments
mkdir -p gpurun_out/r3n; O=$PWD/gpurun_out/r3n; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests/test_gpu_quad_parity.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_quad.log; cat $O/pytest_quad.log
timeout 900 python bench.py --config 4 --no-cpu-baseline --no-host-rate --no-pmc --steps 40 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); r=d['roofline']; k=d['config']; print('config 4 value', d['value'], 'ms', d['ms_per_step'], 'validated', k['converged'], '/', k['instances'], 'iters', k['mean_iterations'], 'passes', k['mean_passes'], 'kernel_ms', r['kernel_ms'])" || tail -5 $O/bench_cfg4.err
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/quad_gpu.py $B > $O/quad_phase_B$B.txt 2>&1; cat $O/quad_phase_B$B.txt | head -24; done
