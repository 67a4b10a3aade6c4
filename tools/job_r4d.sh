#!/bin/bash
# round 4, job D: the block part of the first trial merged into the direction phase (steps kept in registers) -- GPU suite, same-box A/B against the build before it, phase clocks
mkdir -p gpurun_out/r4d
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4d; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 bash tools/ab.sh libobca_hip_prev.so libobca_hip.so 2>&1 | tee $O/ab_sync.txt
for rep in 1 2; do for L in libobca_hip_prev.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_pipe_$L.json 2> $O/bench_pipe_$L.err
  python -c "import json;d=json.loads(open('$O/bench_pipe_$L.json').read().strip().splitlines()[-1]);print('$L pipelined', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged'])" | tee -a $O/ab_pipelined.txt
done; done
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; cat $O/phase_B$B.txt; done
