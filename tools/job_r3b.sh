#!/bin/bash
# round 3, job b: fused line search -- parity suite, bench, phase clocks
mkdir -p gpurun_out/r3b; O=$PWD/gpurun_out/r3b; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_parity.log; cat $O/pytest_parity.log
timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --streams 1 > $O/bench_s1.json 2>/dev/null; cut -c1-200 $O/bench_s1.json
for B in 64 1024; do OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; cat $O/phase_B$B.txt; done
