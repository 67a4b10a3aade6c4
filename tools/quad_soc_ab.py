"""quadcopter kernel: time of a synchronous batch solve with the default options, with IPOPT's second-order correction, with its least-squares initial multipliers, with its
gradient-based objective scaling, and with all three (obca_quadcopter_reference_opts);
OBCA_HIP_LIBRARY selects the library (same-box A/B against the previous build)"""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import obca_amd
from obca_amd import scenarios as S
from obca_amd.api import QuadBatch, Context

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 60; steps = 6
bt = S.make_quad_batch(B, N, random_endpoints=True)
qb = QuadBatch(Context(0), B, N)
qb.upload(bt["x0"], bt["xF"], bt["Ts"], bt["R"], bt["ob"], bt["xWS"], bt["timeWS"])
res = {}
for name in ("default", "max_soc4", "lsq_init", "obj_scaling", "ipopt"):
    o = None
    if name != "default":
        o = obca_amd.quadcopter_ipopt_opts()      # max_soc = 4, lsq_init = 1, obj_scaling = 1
        if name == "max_soc4": o.lsq_init = 0; o.obj_scaling = 0
        if name == "lsq_init": o.max_soc = 0; o.obj_scaling = 0
        if name == "obj_scaling": o.max_soc = 0; o.lsq_init = 0
    try:
        qb.solve(o)
    except obca_amd.ObcaError as e:
        res[name] = str(e); continue
    ms = []
    for _ in range(steps):
        qb.solve(o); ms.append(qb.kernel_ms())
    out = qb.download()
    res[name] = dict(kernel_ms=float(np.median(ms)), solved=int((out["exitflag"] == 1).sum()), iters=int(out["iters"].sum()), nreg=int(out["info"][:, 6].sum()),
                     solves_per_s=float((out["exitflag"] == 1).sum() / (np.median(ms) * 1e-3)))
print(json.dumps(dict(B=B, N=N, lib=os.environ.get("OBCA_HIP_LIBRARY", "libobca_hip.so"), res=res)))
