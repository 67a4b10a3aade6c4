#!/bin/bash
# round 3, job e: parity (incl. two-launch bit identity with the parked assembly), quad suite, bench, phase clocks
mkdir -p gpurun_out/r3e; O=$PWD/gpurun_out/r3e; R=$PWD; C=$R/obca_amd/csrc
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 300 python bench.py --no-cpu-baseline --config 4 --steps 24 > $O/bench4.json 2> $O/bench4.err; cut -c1-200 $O/bench4.json
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; cat $O/phase_B$B.txt; done
