#!/bin/bash
# round 4, job B: the sectioned stage assembly (condensed obstacle sums in LDS, sparse bicycle derivatives, scalar registers for uniform values, Riccati operands in the
# dynamic LDS union) -- GPU suite on the default build, then same-box A/B of the round-3 binary, the default build (one wavefront per SIMD) and the two-wavefronts-per-SIMD build
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4b; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 bash tools/ab.sh libobca_hip_base.so libobca_hip.so libobca_hip_w2.so 2>&1 | tee $O/ab_sync.txt
for L in libobca_hip.so libobca_hip_w2.so; do
  for S in 4 8; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 200 --streams $S --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_pipe_${S}_$L.json 2> $O/bench_pipe_${S}_$L.err
  python -c "import json;d=json.loads(open('$O/bench_pipe_${S}_$L.json').read().strip().splitlines()[-1]);print('$L streams $S pipelined', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged'])" | tee -a $O/ab_pipelined.txt
  done
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 40 --batch 8192 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_B8192_$L.json 2> $O/bench_B8192_$L.err
  python -c "import json;d=json.loads(open('$O/bench_B8192_$L.json').read().strip().splitlines()[-1]);print('$L B=8192', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged'])" | tee -a $O/ab_pipelined.txt
done
