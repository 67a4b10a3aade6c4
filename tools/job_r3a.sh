#!/bin/bash
# round 3, job a: what one-wavefront workgroups can stream (MALL knee, memory-level parallelism), baseline bench on this box, phase clocks against the resident batch
mkdir -p gpurun_out/r3a; O=$PWD/gpurun_out/r3a; R=$PWD
timeout 300 tools/micro/mall_knee > $O/mall_knee.txt 2>&1; cat $O/mall_knee.txt
timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
for B in 256 512 768 1024; do OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; head -2 $O/phase_B$B.txt; done
