#!/bin/bash
# round 2, job j: host-pointer path after the smaller upload (x,u,t only) -- parity of that path, then the PCIe-inclusive rate over 4 / 6 / 8 worker lanes
mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for S in 4 6 8; do
  OBCA_SLOTS=$S timeout 600 python tools/pcie_rate.py 4096 16384 > $O/pcie_s$S.json 2> $O/pcie_s$S.err; cut -c1-1500 $O/pcie_s$S.json
done
