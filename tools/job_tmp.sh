#!/bin/bash
mkdir -p gpurun_out/r3v; O=$PWD/gpurun_out/r3v; C=$PWD/obca_amd/csrc
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest.log
for L in libobca_hip.so libobca_hip_old.so libobca_hip.so libobca_hip_old.so; do for CF in 2 4; do OBCA_HIP_LIBRARY=$C/$L timeout 600 python bench.py --config $CF --no-cpu-baseline --no-host-rate --no-pmc --no-distinct --steps $([ $CF = 2 ] && echo 200 || echo 40) 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L config $CF value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'validated', d['config']['converged'])"; done; done | tee $O/ab.txt
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B 2>&1 | grep -v "^ric_p\|^init" > $O/phase_B$B.txt; cat $O/phase_B$B.txt; done
