#!/bin/bash
# round 5, job D: look for a box like job B's (results of full-machine batches differ between runs) and, on it, find out what differs: arithmetic / LDS / register / HBM
# exchange probes on every CU (tools/micro/cu_consistency), the solver with the unit each instance ran on (tools/determinism_hw.py), NaN-poisoned buffers, the round-4 library
T=$1; mkdir -p gpurun_out/r5d
O=$PWD/gpurun_out/r5d; C=$PWD/obca_amd/csrc
# (job B's failures began ~20 s into a sustained load: the probes run long enough to bring the GPU to its working temperature and power state)
( cd tools/micro && timeout 200 ./cu_consistency 400 ) > $O/cu_$T.txt 2>&1
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 60 > $O/hw_$T.txt 2>&1
grep -E "Uuid: +GPU" $O/hw_$T.txt; tail -n 1 $O/cu_$T.txt; tail -n 1 $O/hw_$T.txt
if ! grep -q "TOTAL differing (instance, run) pairs 0" $O/hw_$T.txt || ! grep -q " 0 deviating" $O/cu_$T.txt; then
  echo "=== box with differing results: deeper probes"
  ( cd tools/micro && timeout 300 ./cu_consistency 16 ) > $O/cu_deep_$T.txt 2>&1; tail -n 12 $O/cu_deep_$T.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 24 > $O/hw_deep_$T.txt 2>&1; grep -E "differing|units of" $O/hw_deep_$T.txt | cut -c1-1500
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python tools/determinism.py 6 > $O/poison_$T.txt 2>&1; tail -n 8 $O/poison_$T.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_r4.so timeout 300 python tools/determinism.py 8 > $O/r4_$T.txt 2>&1; tail -n 8 $O/r4_$T.txt
  ( cd tools/micro && timeout 120 ./lds_atomic_order ) > $O/lds_atomic_$T.txt 2>&1; tail -n 4 $O/lds_atomic_$T.txt
fi
