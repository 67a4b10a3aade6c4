#!/bin/bash
# round 5, job AD (last seconds): the final library: smoke + a minimal bench line
mkdir -p gpurun_out/r5ad
timeout 20 python bench.py --steps 60 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-other-configs --no-ipopt-leg > gpurun_out/r5ad/b.json 2> gpurun_out/r5ad/b.err
python -c "import json;d=json.loads(open('gpurun_out/r5ad/b.json').read().strip().splitlines()[-1]);print('final library: value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['config']['converged'], d['config']['copies_bit_identical'])"
