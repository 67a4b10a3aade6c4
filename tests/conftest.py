import os
import sys
import subprocess
import numpy as np
import pytest

# No byte-code caches of the test modules (pytest's assertion rewriting honours this): the snapshot that travels to the GPU box stays free of derived files, and `gpurun` -- which
# refuses snapshots that carry sanitizer build lines, also inside a .pyc -- never meets the CPU-only sanitizer tests' flags there (tests/emu_sanitize_variants.py, .gpurunignore).
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite: the parity tests of the SURVEY section-8 rows (HIP path against the oracle / the dense pins) run FIRST -- within them the ones on small batches
# before the ones that fill the machine --, the determinism tests next, the bit-equality tests of the plumbing (chunked host calls, multi-device contexts, ranks) after them, the
# tests that leave foreign patterns on the CUs (written at the very end of round 5, when no GPU time was left to run them) LAST:
# `pytest -x` on the driver's box then stops, if it stops, with the parity evidence already on the record.  The sort is stable: ties keep the order written.
_GPU_FILE_ORDER = {"test_gpu_parity.py": 0, "test_gpu_quad_parity.py": 1, "test_gpu_determinism.py": 3, "test_gpu_multi.py": 4, "test_gpu_history.py": 5}
# tests whose batches fill the GPU (>= 640 one-wavefront workgroups): after every small-batch parity test of both kernels
_GPU_LARGE = ("bench_batch", "benchmark_batch", "full_size", "two_launch", "batches_in_flight", "bench_size", "all_1024", "config5_with_binding")


def _gpu_key(item):
    f = os.path.basename(str(item.fspath))
    k = _GPU_FILE_ORDER.get(f, 9)
    if k <= 1 and any(t in item.name for t in _GPU_LARGE):
        k = 2
    return k


def pytest_collection_modifyitems(config, items):
    items.sort(key=_gpu_key)


_SELFTEST = {}


def pytest_sessionfinish(session, exitstatus):
    """after a GPU run: same inputs, same bits?  (obca_amd.selftest: the 1 024-instance bench batch solved four times and once more after a foreign pattern was left in LDS.)  The line
    goes into the terminal summary, so that a failed bit-equality test can be read beside it."""
    if not any(it.get_closest_marker("gpu") for it in session.items):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        import obca_amd
        _SELFTEST.update(obca_amd.selftest(0, repeats=4))
        try:
            _SELFTEST["device"] += ", uuid %s" % torch.cuda.get_device_properties(0).uuid
        except Exception:      # noqa: BLE001 -- older torch: no uuid
            pass
    except Exception as e:      # noqa: BLE001 -- a diagnostic must not change the outcome of the run
        _SELFTEST.update(error=repr(e))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _SELFTEST:
        if "error" in _SELFTEST:
            terminalreporter.write_line("obca self-test of this GPU could not run: " + _SELFTEST["error"])
        else:
            terminalreporter.write_line("obca self-test of this GPU (%s): %d runs of the %d-instance bench batch, %d solved, %d (instance, run) results differ from the first run, "
                                        "%d differ after a foreign pattern was left in the CUs' LDS%s"
                                        % (_SELFTEST["device"], _SELFTEST["runs"], _SELFTEST["instances"], _SELFTEST["solved"], _SELFTEST["differing"], _SELFTEST.get("after_pattern", -1),
                                           "" if _SELFTEST["differing"] == 0 and _SELFTEST.get("after_pattern", 0) == 0 else "  <-- SAME INPUTS, OTHER BITS (DESIGN.md section 11)"))


def gpu_verdict():
    """one line for the message of a failed bit-equality assertion: what the GPU the test ran on does with identical inputs (obca_amd.selftest, ~0.1 s)"""
    try:
        import obca_amd
        r = obca_amd.selftest(0, repeats=3)
        return " [self-test of this GPU right after the failure: %d (instance, run) results of %d x %d differ from the first run, %d after a foreign pattern in LDS%s]" % (
            r["differing"], r["runs"] - 1, r["instances"], r["after_pattern"], "" if r["differing"] == 0 and r["after_pattern"] == 0 else ": SAME INPUTS, OTHER BITS -- DESIGN.md section 11")
    except Exception as e:      # noqa: BLE001
        return " [self-test could not run: %r]" % e


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu():
    """host emulation of the HIP solver (tests/emu/obca_emu.cpp): kernel logic on the CPU, test-only."""
    import emu_solver
    return emu_solver.load()      # (built aside and renamed into place: xdist workers may build at once)


@pytest.fixture(scope="session")
def backwards():
    from obca_amd import scenarios as S
    A, b, v = S.scenario_hrep(S.BACKWARDS)
    return dict(sc=S.BACKWARDS, A=A, b=b, vOb=v, L=S.L_WHEELBASE, ego=S.EGO, XYb=S.XYBOUNDS)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=True)
