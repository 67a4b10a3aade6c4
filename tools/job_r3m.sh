#!/bin/bash
# round 3, job m: backward sweep with one column of the stage system per lane (A/B against the three-phase sweep of commit c3f5d6a)
mkdir -p gpurun_out/r3m; O=$PWD/gpurun_out/r3m; R=$PWD; C=$R/obca_amd/csrc
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config3_bench" 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-host-rate --steps 200 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('value', d['value'], 'traffic', r['traffic'], 'kernel_ms', r['kernel_ms'], r.get('traffic_over_algorithmic'), 'sync', d['config']['single_batch_sync_solves_per_s'])"
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; grep -v "^ric_p\|^init" $O/phase_B$B.txt; done
if [ -f $C/libobca_hip_old.so ]; then OBCA_HIP_LIBRARY=$C/libobca_hip_old.so timeout 600 python bench.py --no-cpu-baseline --no-host-rate --no-pmc --steps 200 > $O/bench_old.json 2> $O/bench_old.err; python -c "
import json; d=json.load(open('$O/bench_old.json')); print('OLD value', d['value'])"; fi
