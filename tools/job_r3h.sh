#!/bin/bash
# round 3, job h: full GPU suite with the full-size config 3 / 5 parity tests; default bench line (live PMC, host-pointer rate, CPU baseline); phase clocks
mkdir -p gpurun_out/r3h; O=$PWD/gpurun_out/r3h; R=$PWD; C=$R/obca_amd/csrc
timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('value', d['value'], 'ms', d['ms_per_step']); print({k:v for k,v in r.items() if k not in ('bound_detail','kernel_timing','pipelined_note','traffic_source','kernel_ms_all')}); print(r['traffic_source']); print(d['cpu_baseline']); print(d['config']['host_pointer'], d['config']['single_batch_sync_solves_per_s'])"; tail -3 $O/bench.err
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; grep -v "^ric_p\|^init" $O/phase_B$B.txt; done
