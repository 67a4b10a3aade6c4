#!/bin/bash
# round 4, job C: GPU suite on the sectioned-assembly build, per-phase clocks (lone instance / full machine), SQ counters, the full bench line
mkdir -p gpurun_out/r4c
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4c; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; cat $O/phase_B$B.txt; done
timeout 900 bash tools/pmc_sq.sh 1024 > $O/pmc_sq.txt 2>&1; cp -r gpurun_out/pmcsq $O/ 2>/dev/null; tail -22 $O/pmc_sq.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
