#!/bin/bash
# round 2, job l: parking kernel with loads-before-stores in the stage back-substitution and block prefetch in the obstacle loops
mkdir -p gpurun_out/r2l
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2l; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest_parking.log; tail -3 $O/pytest_parking.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json
timeout 900 python bench.py --config 5 --no-cpu-baseline --steps 24 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cut -c1-250 $O/bench_cfg5.json
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/phase_profile.py 1024 > $O/phase_B1024.txt; cat $O/phase_B1024.txt
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/phase_profile.py 64 > $O/phase_B64.txt; head -3 $O/phase_B64.txt
