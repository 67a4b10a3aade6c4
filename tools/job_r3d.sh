#!/bin/bash
# round 3, job d: full GPU suite + bench after the explicit-fma seam
mkdir -p gpurun_out/r3d; O=$PWD/gpurun_out/r3d; R=$PWD; C=$R/obca_amd/csrc
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 100 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
