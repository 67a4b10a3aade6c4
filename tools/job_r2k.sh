#!/bin/bash
# round 2, job k: quadcopter kernel with the delta_w = 0 shortcut (q_block_bad): parity, bench line, phase profile
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2k; R=$PWD
timeout 900 python -m pytest tests/test_gpu_quad_parity.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest_quad.log; cat $O/pytest_quad.log
timeout 900 python bench.py --config 4 --no-cpu-baseline --steps 24 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-400 $O/bench_cfg4.json
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 64 > $O/quad_phase_B64.txt; OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 1024 > $O/quad_phase_B1024.txt
cat $O/quad_phase_B1024.txt
