#!/bin/bash
# round 5, job H: soak -- does a box start to differ from itself under sustained load?
mkdir -p gpurun_out/r5h
O=$PWD/gpurun_out/r5h
rocminfo | grep -E "Uuid: +GPU" > $O/uuid.txt; cat $O/uuid.txt
timeout 600 python tools/determinism_soak.py ${1:-240} ipopt > $O/soak.txt 2>&1; tail -n 40 $O/soak.txt | cut -c1-300
