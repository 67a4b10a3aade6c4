#!/bin/bash
# round 3, job c: fused line search + LDS forward sweep + LDS driver state; 4 vs 6 instances per CU
mkdir -p gpurun_out/r3c; O=$PWD/gpurun_out/r3c; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest_parity.log; cat $O/pytest_parity.log
for V in hip hip_w2; do
  OBCA_HIP_LIBRARY=$C/libobca_$V.so timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench_$V.json 2> $O/bench_$V.err; cut -c1-200 $O/bench_$V.json; tail -2 $O/bench_$V.err
  OBCA_HIP_LIBRARY=$C/libobca_$V.so timeout 300 python bench.py --no-cpu-baseline --steps 120 --streams 8 2>/dev/null | cut -c1-200
  OBCA_HIP_LIBRARY=$C/libobca_$V.so timeout 300 python bench.py --no-cpu-baseline --steps 20 --streams 1 2>/dev/null | cut -c1-200
  for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_${V}_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_${V}_B$B.txt 2>&1; cat $O/phase_${V}_B$B.txt; done
done
OBCA_HIP_LIBRARY=$C/libobca_hip_w2_prof.so timeout 200 python tools/phase_profile.py 1536 > $O/phase_hip_w2_B1536.txt 2>&1; cat $O/phase_hip_w2_B1536.txt
