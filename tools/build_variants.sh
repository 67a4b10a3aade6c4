#!/bin/bash
# diagnostic builds of libobca_hip.so, loaded through OBCA_HIP_LIBRARY (git-ignored, not part of the product):
#   poison  -DOBCA_POISON [-DOBCA_POISON_VALUE]   work buffers and the kernels' LDS filled with a pattern at entry: NaN (default) AND 1e30 -- NaN hides behind fmax
# (the -DOBCA_HWID and -DOBCA_DRAIN builds of the round-5 search are gone with the switches they served: docs/HISTORY.md, "Round 5")
R=$(cd "$(dirname "$0")/.." && pwd); V=$R/obca_amd/csrc/variants; mkdir -p $V
cd $R; HIPCC=$(python -m obca_amd.buildflags hipcc)      # warnings are errors in the variants too
$HIPCC -DOBCA_POISON -o $V/libobca_hip_poison.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -DOBCA_POISON -DOBCA_POISON_VALUE=1e30 -o $V/libobca_hip_poison_1e30.so $R/obca_amd/csrc/obca_hip.hip &
wait; ls -la $V
