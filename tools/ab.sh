#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab.sh NEW.so OLD.so [NEW_prof.so OLD_prof.so]   (paths relative to obca_amd/csrc; alternating runs of the lean default line,
# then -- if profiling builds are given -- the per-phase clocks of both, and a bit-for-bit comparison of the downloads of the config-2 bench batch under both option sets)
C=$PWD/obca_amd/csrc; NEW=${1:-libobca_hip.so}; OLD=${2:-variants/libobca_hip_prev.so}
LEAN="--no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs"
for r in 1 2 3; do for V in $NEW $OLD; do OBCA_HIP_LIBRARY=$C/$V timeout 300 python bench.py --steps 60 --warmup 12 $LEAN 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', 'value', d['value'], 'lone launch ms', d['roofline']['kernel_ms'])"; done; done
if [ -n "$3" ]; then for V in $3 $4; do echo "== $V"; OBCA_HIP_LIBRARY=$C/$V timeout 200 python tools/phase_profile.py 1024 ipopt 2>&1 | grep -E "kernel ms|cycles per pass|^other|^ric_bwd|^border_cl|^fwd_seq|^bs_|^trial|^apply|^asm"; done; fi
python - "$C/$NEW" "$C/$OLD" <<'PY'
import os, sys, subprocess, numpy as np
code = r'''
import sys, numpy as np
import obca_amd as OA
from obca_amd import scenarios as S
bt = S.make_batch(S.BACKWARDS, 1024, 80); xWS = bt["xWS"].copy(); xWS[:, 0, :] = bt["x0"]
res = {}
for name, o in (("ref", OA.ipopt_opts()), ("fast", None)):
    out = OA.parking_signed_dist_batch(bt["x0"], bt["xF"], 80, bt["Ts"], bt["L"], bt["ego"], bt["XYbounds"], bt["vOb"], bt["A"], bt["b"], xWS[:, :, 0], xWS[:, :, 1], xWS[:, :, 2], 0, xWS, bt["uWS"], opts=o)
    res[name + "_info"] = out["info"]; res[name + "_xp"] = np.asarray(out["xp"])
np.savez(sys.argv[1], **res)
'''
files = []
for lib in sys.argv[1:3]:
    f = "/tmp/ab_%s.npz" % os.path.basename(lib); files.append(f)
    subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, OBCA_HIP_LIBRARY=lib), check=True)
a, b = np.load(files[0]), np.load(files[1])
print("bit-identical downloads of the config-2 batch under both option sets:", all(np.array_equal(a[k], b[k]) for k in a.files))
PY
