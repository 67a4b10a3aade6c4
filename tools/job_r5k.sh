#!/bin/bash
# round 5, job K: the last binary (SL_XPASS cumulative and kept at the end of the LDS block, corridor test reframed): reproducibility probes, GPU suite, smoke, the driver's 20-step bench line
mkdir -p gpurun_out/r5k
O=$PWD/gpurun_out/r5k; C=$PWD/obca_amd/csrc
rocminfo | grep -E "Uuid: +GPU" > $O/uuid.txt; cat $O/uuid.txt
( cd tools/micro && timeout 200 ./cu_consistency 100 ) > $O/cu.txt 2>&1; tail -n 1 $O/cu.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 40 > $O/hw.txt 2>&1; tail -n 1 $O/hw.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -16 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err
python -c "
import json
d = json.loads(open('gpurun_out/r5k/bench_driver_line.json').read().strip().splitlines()[-1]); k = d['config']; r = d['roofline']
print('driver line: value', d['value'], 'ms', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'validated', k['converged'], '/', k['instances'], 'frac', r['frac'], 'bit-identical', k['copies_bit_identical'], 'fast', k['fast_options']['solves_per_s'], 'other', [(o['config'], o['solves_per_s'], o['validated']) for o in k['other_configs']], 'cpu', d['cpu_baseline']['value'])
"
