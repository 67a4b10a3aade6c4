#!/bin/bash
# round 2, job d: one-wavefront-per-instance parking kernel -- parity suite, bench line, phase profile
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
O=gpurun_out/r2d
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('value',d['value'],'ms',d['ms_per_step'],'conv',d['config']['converged'],'kernel_ms',r['kernel_ms'],'frac',r['frac'],'pipe_frac',r['pipelined_frac'],'launches',r.get('ipm_launches_per_step'))"; tail -3 $O/bench.err
OBCA_HIP_LIBRARY=$PWD/obca_amd/csrc/libobca_hip_prof.so timeout 300 python tools/phase_profile.py 64 > $O/phase_B64.txt 2>&1; cat $O/phase_B64.txt
OBCA_HIP_LIBRARY=$PWD/obca_amd/csrc/libobca_hip_prof.so timeout 300 python tools/phase_profile.py 1024 > $O/phase_B1024.txt 2>&1; cat $O/phase_B1024.txt
