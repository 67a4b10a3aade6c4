#!/bin/bash
# round 5, job L: same-box A/B of the library before / after the SL_XPASS change (one more int in the driver's LDS state)
mkdir -p gpurun_out/r5l
O=$PWD/gpurun_out/r5l; C=$PWD/obca_amd/csrc
for rep in 1 2 3; do for L in variants/libobca_hip_before_xpass.so libobca_hip.so variants/libobca_hip_align16.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-other-configs > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$L value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'fast', d['config']['fast_options']['solves_per_s'], d['config']['converged'])" | tee -a $O/ab_lds_alignment.txt
done; done
