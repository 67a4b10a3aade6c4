#!/bin/bash
# diagnostic builds of libobca_hip.so that the job scripts of round 5 load through OBCA_HIP_LIBRARY (git-ignored, not part of the product):
#   hwid    -DOBCA_HWID                          HW_ID / XCC_ID of the unit every instance ran on (tools/determinism_hw.py)
#   poison  -DOBCA_POISON [-DOBCA_POISON_VALUE]   work buffers and the kernels' LDS filled with a pattern at entry: NaN (default) AND 1e30 -- NaN hides behind fmax
#   drain   -DOBCA_DRAIN                         s_waitcnt vmcnt(0) at every synchronisation point of the parking kernel
R=$(cd "$(dirname "$0")/.." && pwd); V=$R/obca_amd/csrc/variants; mkdir -p $V
cd $R; HIPCC=$(python -m obca_amd.buildflags hipcc)      # warnings are errors in the variants too
$HIPCC -DOBCA_HWID -o $V/libobca_hip_hwid.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -DOBCA_POISON -o $V/libobca_hip_poison.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -DOBCA_POISON -DOBCA_POISON_VALUE=1e30 -o $V/libobca_hip_poison_1e30.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -DOBCA_DRAIN -o $V/libobca_hip_drain.so $R/obca_amd/csrc/obca_hip.hip &
wait; ls -la $V
