#!/bin/bash
# round 4, job J: the driver's invocation times 20 steps: how much of that short region is ramp-up / drain, and does the number of batches in flight change it
mkdir -p gpurun_out/r4j
O=$PWD/gpurun_out/r4j
for rep in 1 2; do for st in 4 6 8 12 16; do
  timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('steps 20 warmup 5 streams $st: value', d['value'], 'ms/step', d['ms_per_step'])" | tee -a $O/short_region.txt
done; done
for st in 8 16; do timeout 120 python bench.py --steps 200 --streams $st --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('steps 200 streams $st: value', d['value'], 'ms/step', d['ms_per_step'])" | tee -a $O/short_region.txt; done
