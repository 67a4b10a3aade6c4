"""Per-function register / scratch usage of the gfx950 code object (hipcc -save-temps assembly): python tools/regs.py [extra hipcc flags...]"""
import os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from obca_amd.buildflags import HIPCC      # the product's flags
d = tempfile.mkdtemp()
cmd = HIPCC + [
       "-save-temps", "-Wno-error",      # (the preprocessed intermediate loses the macro provenance some warnings are silenced by)
       "-o", "t.so", R + "/obca_amd/csrc/obca_hip.hip"] + sys.argv[1:]
subprocess.run(cmd, cwd=d, stderr=subprocess.DEVNULL, check=True)
s = open(os.path.join(d, "obca_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
name = None; rows = []
for ln in s.splitlines():
    m = re.match(r"^(_Z\w+):", ln)
    if m: name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]; cur = {}
    for k in ("codeLenInByte", "NumVgprs", "NumAgprs", "ScratchSize", "Occupancy"):
        m = re.match(r"^; %s[:=]? *=? *(\d+)" % k, ln)
        if m and name: cur[k] = int(m.group(1))
    if ln.startswith("; ScratchSize") and name: rows.append((name, dict(cur)))
sp = {}
for m in re.finditer(r"\.name:\s+(\S+)|\.vgpr_spill_count:\s+(\d+)|\.private_segment_fixed_size:\s+(\d+)|\.vgpr_count:\s+(\d+)", s): pass
for n, c in rows: print("%-70s code %6d  vgpr %3d  agpr %3d  scratch %5d" % (n[-70:], c.get("codeLenInByte", 0), c.get("NumVgprs", 0), c.get("NumAgprs", 0), c.get("ScratchSize", 0)))
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", s, re.S):
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    print("KERNEL %-50s vgpr_count %s spill %s private %s lds %s" % (subprocess.run(["c++filt", g("name").group(1)], capture_output=True, text=True).stdout.strip().split("(")[0][:50], g("vgpr_count").group(1), g("vgpr_spill_count").group(1), g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1)))
