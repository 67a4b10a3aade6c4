#!/bin/bash
# round 5, job C: which hardware unit produces the results that differ between runs?  (job B's box: 11 GPU tests with >= 640 instances per batch failed on bit equality, job A's box: none)
mkdir -p gpurun_out/r5c
O=$PWD/gpurun_out/r5c; C=$PWD/obca_amd/csrc
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 600 python tools/determinism_hw.py 8 > $O/determinism_hw.txt 2>&1; grep -v "^  config\|^  smaller\|^  two" $O/determinism_hw.txt | tail -n 40
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_r4.so timeout 300 python tools/determinism.py 10 > $O/determinism_r4.txt 2>&1; tail -n 3 $O/determinism_r4.txt
OBCA_HIP_LIBRARY=$C/libobca_hip.so timeout 300 python tools/determinism.py 10 > $O/determinism_new.txt 2>&1; tail -n 3 $O/determinism_new.txt
