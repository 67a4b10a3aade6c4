#!/bin/bash
# round 5, job N: do results differ right after the GPU wakes from its low-power state, with or without CPU load?  Then a pytest-like pattern: the GPU suite's first 16 tests twice.
mkdir -p gpurun_out/r5n
O=$PWD/gpurun_out/r5n
rocminfo | grep -E "Uuid: +GPU" | tee $O/uuid.txt
timeout 400 python tools/determinism_idle_burst.py 200 > $O/idle_burst.txt 2>&1; tail -n 8 $O/idle_burst.txt | cut -c1-300
