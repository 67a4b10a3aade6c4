#!/bin/bash
# builds libobca_hip.so (product), the -DOBCA_PROFILE diagnostic variant, the oracle and the host emulation
R=$(cd "$(dirname "$0")/.." && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-optimize-sibling-calls -I$R/include"
hipcc $F -o $R/obca_amd/csrc/libobca_hip.so $R/obca_amd/csrc/obca_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" | head -5 &
hipcc $F -DOBCA_PROFILE -o $R/obca_amd/csrc/libobca_hip_prof.so $R/obca_amd/csrc/obca_hip.hip 2>&1 | grep -E "error" &
g++ -O1 -std=c++17 -fPIC -shared -Wno-unknown-pragmas -o $R/tests/emu/libobca_emu.so $R/tests/emu/obca_emu.cpp 2>&1 | grep -E "error" -A3 &
make -C $R/oracle -s &
g++ -O2 -std=c++17 -shared -fPIC -pthread -o $R/obca_amd/csrc/libobca_plan.so $R/obca_amd/csrc/obca_planner.cpp $R/obca_amd/csrc/obca_planner_ref.cpp &
wait
