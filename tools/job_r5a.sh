#!/bin/bash
# round 5, job A: the non-determinism the driver's round-4 GPU run met (test_chunked_host_call_equals_resident_batch[1000-2]).
#   1. micro probe: is one ds_add_f64 with several lanes on one address served in lane order, alone and under load?
#   2. tools/determinism_ragged.py on the round-4 library (ds_add_f64 sums), on the round-4 solver with NaN-poisoned work buffers / LDS, on the new library (ordered sums,
#      initialised allocations) and on its poisoned build
#   3. the GPU suite on the new library, a same-box A/B of the two libraries on the bench batch
mkdir -p gpurun_out/r5a
O=$PWD/gpurun_out/r5a; C=$PWD/obca_amd/csrc
( cd tools/micro && timeout 120 ./lds_atomic_order ) > $O/lds_atomic_order.txt 2>&1
for L in variants/libobca_hip_r4.so variants/libobca_hip_r4poison.so libobca_hip.so variants/libobca_hip_poison.so; do
  R=40; case $L in *poison*) R=12;; esac
  OBCA_HIP_LIBRARY=$C/$L timeout 400 python tools/determinism_ragged.py $R 40 150 > $O/ragged_$(basename $L .so).txt 2>&1
done
OBCA_HIP_LIBRARY=$C/libobca_hip.so timeout 300 python tools/determinism_ragged.py 16 40 150 reference > $O/ragged_libobca_hip_reference_opts.txt 2>&1
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_r4.so timeout 300 python tools/determinism_ragged.py 16 40 150 reference > $O/ragged_libobca_hip_r4_reference_opts.txt 2>&1
tail -n 3 $O/ragged_*.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -n 5 $O/pytest_gpu.log
for rep in 1 2; do for L in variants/libobca_hip_r4.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b.json 2> $O/b.err
  python -c "import json;d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]);print('$L pipelined', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'sync', d['config']['single_batch_sync_solves_per_s'], d['config']['converged'])" | tee -a $O/ab_ordered_sum.txt
done; done
for L in variants/libobca_hip_r4.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --config 5 --steps 30 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/b5.json 2> $O/b5.err
  python -c "import json;d=json.loads(open('$O/b5.json').read().strip().splitlines()[-1]);print('$L config5 pipelined', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d['config'].get('converged'))" | tee -a $O/ab_ordered_sum.txt
done
