#!/bin/bash
# A/B of library builds on the same GPU box: tools/ab.sh libA.so libB.so ...  (kernel ms of the config-2 batch, 3 repetitions each, interleaved)
for rep in 1 2 3; do for L in "$@"; do
  echo -n "$L  "; OBCA_HIP_LIBRARY=$PWD/obca_amd/csrc/$L timeout 300 python bench.py --steps 5 --warmup 1 --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'], d['value'])"
done; done
