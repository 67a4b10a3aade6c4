#!/bin/bash
# builds libobca_hip.so (product), the -DOBCA_PROFILE diagnostic variant, the oracle, the host emulation and the planner.  The flags -- warnings are errors -- live in
# obca_amd/buildflags.py; the compiler's diagnostics are NOT filtered (rounds 1-5 piped them through grep and missed "variable 'sumz' set but not used").
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
HIPCC=$(python -m obca_amd.buildflags hipcc); GXX=$(python -m obca_amd.buildflags gxx)
$HIPCC -o $R/obca_amd/csrc/libobca_hip.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -DOBCA_PROFILE -o $R/obca_amd/csrc/libobca_hip_prof.so $R/obca_amd/csrc/obca_hip.hip &
$HIPCC -o $R/obca_amd/csrc/libobca_diag.so $R/obca_amd/csrc/obca_diag.hip &
$GXX -O1 -o $R/tests/emu/libobca_emu.so $R/tests/emu/obca_emu.cpp &
make -C $R/oracle -s &
$GXX -O2 -pthread -I$R/include -o $R/obca_amd/csrc/libobca_plan.so $R/obca_amd/csrc/obca_planner.cpp $R/obca_amd/csrc/obca_planner_ref.cpp &
wait
# register / scratch usage per kernel: python tools/regs.py
