#!/bin/bash
# round 2, job a: GPU suite after the host-side rework (chunked / multi-device data path) + PCIe-inclusive rates
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a/pytest_gpu.log; cat gpurun_out/r2a/pytest_gpu.log
timeout 300 python tools/pcie_rate.py 1024 4096 > gpurun_out/r2a/pcie_default.json 2> gpurun_out/r2a/pcie.err; cat gpurun_out/r2a/pcie_default.json; tail -3 gpurun_out/r2a/pcie.err
OBCA_SLOTS=4 OBCA_CHUNK=256 timeout 300 python tools/pcie_rate.py 1024 4096 > gpurun_out/r2a/pcie_s4c256.json 2>> gpurun_out/r2a/pcie.err; cat gpurun_out/r2a/pcie_s4c256.json
OBCA_SLOTS=4 OBCA_CHUNK=1024 timeout 300 python tools/pcie_rate.py 4096 16384 > gpurun_out/r2a/pcie_s4c1024.json 2>> gpurun_out/r2a/pcie.err; cat gpurun_out/r2a/pcie_s4c1024.json
OBCA_SLOTS=2 OBCA_CHUNK=512 timeout 300 python tools/pcie_rate.py 1024 4096 > gpurun_out/r2a/pcie_s2c512.json 2>> gpurun_out/r2a/pcie.err; cat gpurun_out/r2a/pcie_s2c512.json
