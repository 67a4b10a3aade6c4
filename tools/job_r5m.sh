#!/bin/bash
# round 5, job M: where the dynamic LDS block starts (padding behind the static block, 0..240 bytes in steps of 16) against `value` -- same box, two rounds
mkdir -p gpurun_out/r5m
O=$PWD/gpurun_out/r5m; C=$PWD/obca_amd/csrc
for rep in 1 2; do for pad in 0 16 32 48 64 80 96 112 128 144 160 176 192 208 224 240; do
  L=variants/libobca_hip_pad$pad.so; [ $pad = 0 ] && L=libobca_hip.so
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-other-configs --no-ipopt-leg > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('pad %3d value' % $pad, d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['config']['converged'])" | tee -a $O/lds_pad_scan.txt
done; done
