#!/bin/bash
# round 5, job J: hunt for a box that does not reproduce its own results (quick check; deeper probes only on such a box)
# does not gets the deeper probes instead (hardware unit of every differing instance, pattern probes per CU, poisoned build, round-4 library)
mkdir -p gpurun_out/r5j_$1
O=$PWD/gpurun_out/r5j_$1; C=$PWD/obca_amd/csrc
rocminfo | grep -E "Uuid: +GPU" > $O/uuid.txt; cat $O/uuid.txt
( cd tools/micro && timeout 200 ./cu_consistency 100 ) > $O/cu.txt 2>&1; tail -n 1 $O/cu.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 40 > $O/hw.txt 2>&1; tail -n 1 $O/hw.txt
if ! grep -q "TOTAL differing (instance, run) pairs 0" $O/hw.txt || ! grep -q " 0 deviating" $O/cu.txt; then
  echo "=== this box does not reproduce its own results: deeper probes instead of the bench"
  ( cd tools/micro && timeout 300 ./cu_consistency 400 ) > $O/cu_deep.txt 2>&1; tail -n 14 $O/cu_deep.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 400 python tools/determinism_hw.py 120 > $O/hw_deep.txt 2>&1; grep -E "differing|units of" $O/hw_deep.txt | cut -c1-2500
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python tools/determinism.py 12 > $O/poison.txt 2>&1; tail -n 8 $O/poison.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_r4.so timeout 300 python tools/determinism.py 12 > $O/r4.txt 2>&1; tail -n 8 $O/r4.txt
  ( cd tools/micro && timeout 120 ./lds_atomic_order ) > $O/lds_atomic.txt 2>&1; tail -n 4 $O/lds_atomic.txt
  timeout 300 python tools/determinism_soak.py 60 ipopt > $O/soak.txt 2>&1; tail -n 12 $O/soak.txt | cut -c1-300
  exit 0
fi
echo "box reproduces its own results"; exit 0
