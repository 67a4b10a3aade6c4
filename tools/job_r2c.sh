#!/bin/bash
# round 2, job c: GPU suite after VMAX 8 / quad N 128 / block restoration; sustained-clock check of the pipelined rate
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
O=gpurun_out/r2c
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for K in 20 80 320; do timeout 300 python bench.py --steps $K --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps',d['steps'],'value',d['value'],'ms',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"; done | tee $O/steps_sweep.txt
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > $O/smi.txt; cat $O/smi.txt | head -20
