#!/bin/bash
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2f; R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/job_variants.sh default lds
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so timeout 300 python tools/phase_profile.py 64 > $O/phase_B64.txt 2>&1; cat $O/phase_B64.txt
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so timeout 300 python tools/phase_profile.py 1024 > $O/phase_B1024.txt 2>&1; cat $O/phase_B1024.txt
