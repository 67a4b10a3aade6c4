"""The C-ABI library loads on a machine without a GPU, exports every symbol include/obca_hip.h declares, and fails loudly."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import obca_amd
    path = obca_amd.build_library()
    assert os.path.exists(path)
    return C.CDLL(path)


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "obca_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef OBCA_PROFILE.*?#endif", "", txt, flags=re.S)      # (declared for the profiling build only: not exported by the product library)
    return sorted(set(re.findall(r"\b(obca_[a-z_0-9]+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), s
    from obca_amd.api import EXPORTS
    assert sorted(EXPORTS) == syms


def test_opts_struct_matches_and_defaults(lib):
    from obca_amd.api import Opts
    o = Opts()
    assert lib.obca_default_opts(C.byref(o)) == 0
    # the reference's IPOPT options (ParkingSignedDist.jl:41-43)
    assert o.tol == 1e-5 and o.max_iter == 200 and o.dw_min == 1e-12 and o.dc_bar == 1e-7
    import oracle as O
    oo = O.default_opts()
    for n, _ in Opts._fields_:
        if True:
            assert getattr(o, n) == getattr(oo, n), n      # every option exists under the same name in the checker
    assert o.max_soc == 0 and o.recalc_y == 0 and o.lsq_init == 0 and o.obj_scaling == 0
    q = Opts(); assert lib.obca_quadcopter_reference_opts(C.byref(q)) == 0
    assert q.max_soc == 4 and q.recalc_y == 0 and q.lsq_init == 1 and q.obj_scaling == 1 and q.max_iter == 3000 and q.dw_min == 1e-10      # QuadcopterSignedDist.jl:28-31 (recalc_y = "no") + IPOPT's defaults max_soc, least-squares y0


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import obca_amd
    with pytest.raises(obca_amd.ObcaError):
        obca_amd.Context(0)
    with pytest.raises(obca_amd.ObcaError):
        obca_amd.DualMultWS(1, 1, [1], [[0, 1]], [5.0], [0, 0], [0, 0], [0, 0], [3.7, 1, 1, 1])


def test_product_does_not_import_oracle():
    # the product package must never reach into oracle/ (or tests/emu)
    pkg = os.path.join(ROOT, "obca_amd")
    for dp_, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dp_, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "obca_oracle" not in src, f
                assert "libobca_emu" not in src, f


def test_no_statement_hides_at_the_end_of_a_comment():
    """From round 4 to the end of round 5 the line `... out.cmax = cmx;      // (largest |s z| ...) out.sumy = sumy; out.sumz = sumz;` kept two stores of the parking kernels'
    assembly inside a comment.  No line of the kernel, host or oracle sources may end a `//` comment with what reads like a statement."""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stmt = re.compile(r"//.*[\)\.;] +[A-Za-z_][A-Za-z_0-9\.\[\]>-]* *[-+*/|&]?= *[^=].*;\s*$|//.*[\)\.;] +(if|for|while|return|break|continue)\b.*[;}]\s*$|//.*[\)\.] +[A-Za-z_][A-Za-z_0-9:]*\(.*\);\s*$")
    files = [f for pat in ("obca_amd/csrc/*.h", "obca_amd/csrc/*.hip", "obca_amd/csrc/*.cpp", "oracle/*.c", "oracle/*.h", "tests/emu/*.cpp", "include/*.h") for f in glob.glob(os.path.join(root, pat))]
    assert len(files) >= 10
    hits = []
    for f in files:
        for n, line in enumerate(open(f, errors="replace"), 1):
            if stmt.search(line):
                hits.append("%s:%d: %s" % (os.path.relpath(f, root), n, line.strip()[-120:]))
    assert not hits, "\n".join(hits)


def test_device_sources_compile_without_a_single_warning():
    """The fence for the bug class of DESIGN.md section 11: `hipcc -Wall` prints the two lost stores of round 4 as "variable 'sumz' set but not used".  The product's flags
    (obca_amd/buildflags.py) make warnings errors; this test runs the front end over the kernel and host sources with those flags (device + host pass, every template the
    kernels instantiate; -fsyntax-only: seconds) for the product and for each build variant that changes kernel code, and wants the compiler to say NOTHING."""
    import subprocess
    from obca_amd.buildflags import HIPCC, WARN
    assert "-Werror" in WARN and "-Wall" in WARN and "-Wextra" in WARN and not any(f.startswith("-Wno-unused-but-set") or f in ("-w", "-Wno-unused-variable", "-Wno-uninitialized") for f in HIPCC)
    src = os.path.join(ROOT, "obca_amd", "csrc", "obca_hip.hip")
    base = [f for f in HIPCC if f not in ("-shared", "-fPIC")] + ["-fsyntax-only", "-Wno-unused-command-line-argument"]
    for variant in ([], ["-DOBCA_PROFILE"], ["-DOBCA_POISON"]):
        r = subprocess.run(base + variant + [src], capture_output=True, text=True)
        assert r.returncode == 0 and not r.stdout.strip() and not r.stderr.strip(), (variant, r.stdout[-2000:], r.stderr[-2000:])


def test_the_warning_fence_sees_a_lost_store(tmp_path):
    """...and the fence works: the round-4 line, with the two stores back inside the comment, does not compile."""
    import shutil, subprocess
    from obca_amd.buildflags import HIPCC
    csrc = tmp_path / "csrc"; shutil.copytree(os.path.join(ROOT, "obca_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("*.so", "variants"))
    f = csrc / "obca_solver_assemble.h"; txt = f.read_text()
    assert txt.count("    out.sumy = sumy; out.sumz = sumz;\n") == 1
    f.write_text(txt.replace("    out.sumy = sumy; out.sumz = sumz;\n", "    // out.sumy = sumy; out.sumz = sumz;\n"))
    r = subprocess.run([x for x in HIPCC if x not in ("-shared", "-fPIC")] + ["-fsyntax-only", "-Wno-unused-command-line-argument", str(csrc / "obca_hip.hip")], capture_output=True, text=True)
    assert r.returncode != 0 and "set but not used" in r.stderr, r.stderr[-1500:]


def test_diagnostics_are_not_in_the_product_library(lib):
    """the pattern kernel lives in libobca_diag.so (include/obca_diag.h), the per-phase clocks in the -DOBCA_PROFILE build: the product library exports neither"""
    import subprocess
    from obca_amd import diag
    for s_ in ("obca_debug_leave_pattern", "obca_batch_debug_phase_cycles", "obca_quad_batch_debug_phase_cycles", "obca_diag_leave_pattern"):
        assert not hasattr(lib, s_), s_
    syms = subprocess.run(["nm", "-D", "--defined-only", lib._name], capture_output=True, text=True).stdout
    assert "dirty_kernel" not in syms and "debug" not in syms
    d = C.CDLL(diag.build_library())
    assert hasattr(d, "obca_diag_leave_pattern")
    d.obca_diag_leave_pattern.argtypes = [C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    assert d.obca_diag_leave_pattern(-1, 4, 1e30, None, None) == -1      # (argument check only: no device is touched)


def test_loading_the_library_asks_for_one_hardware_queue_per_stream():
    """obca_amd.api._load() sets GPU_MAX_HW_QUEUES=16 before the HIP runtime's first call unless the caller set a value (the runtime's default of 4 serialises
    the streams that share a queue: profiles/r06_hw_queues.txt); a value the caller has set stays."""
    import subprocess, sys
    code = ("import os, sys; sys.path.insert(0, %r); pre = os.environ.get('GPU_MAX_HW_QUEUES'); "
            "from obca_amd import api; api._load(); print(pre, os.environ.get('GPU_MAX_HW_QUEUES'))") % ROOT
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["None", "16"], out
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="6"), capture_output=True, text=True, check=True).stdout.split()
    assert out == ["6", "6"], out
