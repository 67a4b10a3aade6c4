import os
import sys
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite: the parity tests of the SURVEY section-8 rows (HIP path against the oracle / the dense pins) run FIRST, the determinism tests next, the
# bit-equality tests of the plumbing (chunked host calls, multi-device contexts, ranks) LAST -- `pytest -x` on the driver's box then stops, if it stops, with the parity
# evidence already on the record.  The sort is stable: the order inside a file is the order written.
_GPU_FILE_ORDER = {"test_gpu_parity.py": 0, "test_gpu_quad_parity.py": 1, "test_gpu_determinism.py": 2, "test_gpu_multi.py": 9}


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: _GPU_FILE_ORDER.get(os.path.basename(str(it.fspath)), 5))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu():
    """host emulation of the HIP solver (tests/emu/obca_emu.cpp): kernel logic on the CPU, test-only."""
    import ctypes as C
    src = os.path.join(ROOT, "tests", "emu", "obca_emu.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libobca_emu.so")
    deps = [src] + [os.path.join(ROOT, "obca_amd", "csrc", f) for f in ("obca_solver.h", "obca_model.h", "obca_quad_solver.h", "obca_quad_model.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="session")
def backwards():
    from obca_amd import scenarios as S
    A, b, v = S.scenario_hrep(S.BACKWARDS)
    return dict(sc=S.BACKWARDS, A=A, b=b, vOb=v, L=S.L_WHEELBASE, ego=S.EGO, XYb=S.XYBOUNDS)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)
