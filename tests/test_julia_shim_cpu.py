"""Julia is not in this image, so julia/*.jl cannot be executed here.  What CAN be checked statically: every `ccall` of the shim names an
exported entry point and its Julia type tuple matches the C prototype in include/obca_hip.h parameter by parameter (count, int / double /
pointer kind), and the quadcopter warm start is passed in the reference's 12 x (N+1) orientation."""
import os
import re
from conftest import ROOT

JL = {"Cint": "int", "Cdouble": "double", "Clonglong": "longlong"}


def c_prototypes():
    txt = open(os.path.join(ROOT, "include", "obca_hip.h")).read() + open(os.path.join(ROOT, "include", "obca_plan.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(obca_[a-z_0-9]+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",")] if m.group(2).strip() not in ("", "void") else []
        kinds = []
        for p in params:
            if "*" in p or "[" in p:
                kinds.append("ptr")
            elif re.match(r"(const\s+)?double\b", p):
                kinds.append("double")
            elif re.match(r"(const\s+)?int\b", p):
                kinds.append("int")
            else:
                kinds.append("?" + p)
        protos[m.group(1)] = kinds
    return protos


def jl_ccalls(path):
    src = open(path).read()
    out = []
    for m in re.finditer(r"ccall\(\(:(obca_[a-z_0-9]+), (?:LIB|PLAN)\),\s*(\w+),\s*\(", src):
        i = m.end(); depth = 1; j = i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0); j += 1
        types = [t.strip() for t in re.split(r",(?![^{]*\})", src[i:j - 1]) if t.strip()]
        kinds = ["ptr" if t.startswith(("Ptr{", "Ref{")) or t == "Cstring" else JL.get(t, "?" + t) for t in types]
        out.append((m.group(1), kinds))
    return out


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    n = 0
    for fn in sorted(os.listdir(os.path.join(ROOT, "julia"))):
        if not fn.endswith(".jl"):
            continue
        for name, kinds in jl_ccalls(os.path.join(ROOT, "julia", fn)):
            assert name in protos, (fn, name)
            assert kinds == protos[name], (fn, name, kinds, protos[name])
            n += 1
    assert n >= 12


def test_quadcopter_warm_start_orientation():
    src = open(os.path.join(ROOT, "julia", "OBCAHip.jl")).read()
    # mainQuadcopter.jl:136 / QuadcopterSignedDist.jl:201: xWS is 12 x (N+1); the shim must slice columns, never transpose it
    assert "f64(xWS)[:, 1:N+1]" in src and "permutedims(f64(xWS)[1:N+1, :])), C_NULL,\n               [Float64(timeWS)]" not in src
    for f in ("QuadcopterSignedDist(", "QuadcopterDist(", "ParkingSignedDist(", "ParkingDist(", "DualMultWS(", "MultiContext("):
        assert f in src, f


def test_julia_options_record_mirrors_the_header():
    """julia/OBCAHip.jl `Opts` must list the fields of `obca_opts` (include/obca_hip.h) in order and kind -- the record is handed to the library by pointer"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "obca_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct obca_opts \{(.*?)\} obca_opts;", hdr, flags=re.S).group(1)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            kind, names = decl.split(None, 1)
            c_fields += [(n.strip(), kind) for n in names.split(",")]
    src = open(os.path.join(ROOT, "julia", "OBCAHip.jl")).read()
    jl = re.search(r"mutable struct Opts\n(.*?)\n\s*Opts\(\) = new\(\)", src, flags=re.S).group(1)
    j_fields = [(m.group(1), JL[m.group(2)]) for m in re.finditer(r"(\w+)::(Cdouble|Cint)", jl)]
    assert j_fields == c_fields and len(c_fields) == 35 and c_fields[-5:] == [("max_soc", "int"), ("recalc_y", "int"), ("lsq_init", "int"), ("obj_scaling", "int"), ("restoration", "int")]
