#!/bin/bash
# round 4, job H: how the pipelined rate of config 2 depends on the batch size and the number of batches in flight (what part of the timed step is per-launch overhead
# and tail, what part is the kernel's own rate with every slot busy)
mkdir -p gpurun_out/r4h
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4h
for cfg in "1024 4" "1024 8" "2048 4" "4096 4" "8192 2" "16384 1"; do set -- $cfg
  timeout 300 python bench.py --batch $1 --streams $2 --steps $((204800 / $1)) --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python -c "import json;d=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1]);print('B $1 streams $2: value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'dualws', d['roofline']['dualws_kernel_ms'], 'passes', d['roofline']['passes_per_launch'], 'launches', d['roofline']['ipm_launches_per_step'])" | tee -a $O/batch_sensitivity.txt
done
