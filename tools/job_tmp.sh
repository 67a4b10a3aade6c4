#!/bin/bash
O=$PWD/gpurun_out/r3final
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.log
timeout 900 python bench.py --config 5 --no-cpu-baseline --no-host-rate --steps 60 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --backend gloo --config 5 --steps 24 --warmup 4 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_2rank_gloo_cfg5.json 2> $O/bench_2rank_gloo_cfg5.err
OBCA_HIP_LIBRARY=$PWD/obca_amd/csrc/libobca_hip_prof.so timeout 300 python tools/phase_profile5.py > $O/phase_config5_B1024.txt 2>&1
python -c "
import json
for c in ('bench_cfg5','bench_2rank_gloo_cfg5'):
    d=json.loads(open('$O/'+c+'.json').read().strip().splitlines()[-1]); print(c, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged'], d['roofline'].get('traffic'))"
