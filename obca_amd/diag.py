"""ctypes front end of libobca_diag.so (include/obca_diag.h, obca_amd/csrc/obca_diag.hip): DIAGNOSTICS in a library of their own -- the product library libobca_hip.so
contains none of it.  Used by tests/test_gpu_history.py, obca_amd.selftest() and the bit-equality line of bench.py."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "obca_diag.hip")
_LIB = os.path.join(_HERE, "csrc", "libobca_diag.so")
_lib = None


def build_library(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(os.path.join(_HERE, "..", "include", "obca_diag.h"))):
        from .buildflags import HIPCC
        subprocess.check_call(HIPCC + ["-o", _LIB, _SRC])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} is missing: build it with obca_amd.diag.build_library() / __graft_entry__.build()")
        _lib = C.CDLL(_LIB)
        _lib.obca_diag_leave_pattern.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def leave_pattern(devices, mask=4, value=1e30):
    """fill what later workgroups inherit on a SIMD / CU -- bit 0 vector registers, 1 accumulation registers, 2 the CUs' LDS (with the double `value`), 3 scratch -- on every
    device of `devices` (an index, a list, or an obca_amd.Context); the solves that follow must return the same bits.  Returns [(units covered, units with >= 4 workgroups)]
    per device: how much of the machine the pattern reached."""
    if hasattr(devices, "devices"):
        devices = devices.devices
    if isinstance(devices, int):
        devices = [devices]
    out = []
    for d in devices:
        a, b = C.c_int(0), C.c_int(0)
        rc = _load().obca_diag_leave_pattern(int(d), int(mask), float(value), C.byref(a), C.byref(b))
        if rc != 0:
            raise RuntimeError(f"obca_diag_leave_pattern(device {d}) failed ({rc})")
        out.append((a.value, b.value))
    return out
