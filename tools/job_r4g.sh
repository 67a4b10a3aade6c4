#!/bin/bash
# round 4, job G: quadcopter kernel with the second-order correction option: the rate of the default option set (the option threads a runtime flag through the
# assembly / back-substitution phases: compare with the earlier builds' 48.0 ms at B = 1024) and with max_soc = 4, and the GPU parity tests of the quadcopter path
mkdir -p gpurun_out/r4g
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4g; R=$PWD; C=$R/obca_amd/csrc
for rep in 1 2; do timeout 300 python tools/quad_soc_ab.py 1024 2>&1 | tail -1 | tee -a $O/ab_quad_soc.txt; done
timeout 300 python tools/quad_soc_ab.py 4096 2>&1 | tail -1 | tee -a $O/ab_quad_soc.txt
timeout 1500 python -m pytest tests/test_gpu_quad_parity.py -q -x -s 2>&1 | tail -30 | tee $O/pytest_quad.log
