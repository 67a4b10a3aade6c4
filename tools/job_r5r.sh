#!/bin/bash
# round 5, job R: NaN patterns left in registers / LDS / scratch by a kernel launched in front of every solve: does the solver read anything it has not written?
mkdir -p gpurun_out/r5r
timeout 600 python tools/determinism_dirty.py > gpurun_out/r5r/dirty.txt 2>&1; cat gpurun_out/r5r/dirty.txt | cut -c1-250
