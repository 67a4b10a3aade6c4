"""ONE place for the compile flags of every native piece (product library, its diagnostic variants, planner, host emulation); `python -m obca_amd.buildflags hipcc|gxx`
prints a flag set for the shell scripts under tools/.

Warnings are errors everywhere.  Round 4 lost two stores of the stage assembly behind a `//` comment; `hipcc -Wall` prints that as "variable 'sumz' set but not used" -- the
build scripts of rounds 1-5 never passed -Wall and filtered the compiler's output, and the bug cost two rounds.  tests/test_abi_cpu.py compiles the
device sources with these flags (-fsyntax-only, seconds) and asserts that the compiler prints nothing.
  -Wno-unused-parameter: phase functions share signatures (dw, dc, ... are passed to every variant, used by some) -- the one warning class that is interface, not accident.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
INCLUDE = os.path.join(_HERE, "..", "include")
WARN = ["-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter"]
# -fno-optimize-sibling-calls: the phases of the solve are non-inlined local device functions that use the whole register file.  LLVM drops the
# callee-saved-register saves of such functions (every caller is known) only if no call site is marked `tail`, and -O3 marks them all; with the
# flag the 112 VGPR + ~150 AGPR saves / restores per phase call disappear: 0.23 MB less scratch traffic per factorisation pass, 140 k -> 154 k solves/s.
HIPCC = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-optimize-sibling-calls", "-I" + INCLUDE] + WARN
# host C++ (planner; the emulation of the kernels under tests/emu, which sees `#pragma unroll`)
GXX = ["g++", "-std=c++17", "-fPIC", "-shared"] + WARN + ["-Wno-unknown-pragmas", "-Wno-misleading-indentation"]

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "hipcc"
    print(" ".join({"hipcc": HIPCC, "gxx": GXX, "warn": WARN}[which]))
