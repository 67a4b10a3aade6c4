#!/bin/bash
# round 4, job F: DualMultWS kernel (2-row class) compiled for four wavefronts per SIMD (128 registers, 35 spilled) against three (158 registers): same-box A/B
mkdir -p gpurun_out/r4f
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4f; R=$PWD; C=$R/obca_amd/csrc
for rep in 1 2; do for L in libobca_hip_prev.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_pipe_$L.json 2> $O/bench_pipe_$L.err
  python -c "import json;d=json.loads(open('$O/bench_pipe_$L.json').read().strip().splitlines()[-1]);print('$L pipelined', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['dualws_kernel_ms'], d['config']['single_batch_sync_solves_per_s'], d['config']['converged'])" | tee -a $O/ab_dualws.txt
done; done
rocm-smi --showclocks --showtemp --showuse 2>&1 | head -30 > $O/rocm_smi.txt; timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "dualws or limits or dense" 2>&1 | tail -40
