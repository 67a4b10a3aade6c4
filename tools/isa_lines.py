"""Which SOURCE LINES the instructions of a phase function come from: python tools/isa_lines.py ph_fused2 [top-level-loop-index] [N]
(loop index past the last loop = the code outside the loops; hipcc -save-temps -gline-tables-only; per source line of obca_solver.h / obca_model.h: instructions by class inside the chosen Depth-1 loop of the function, largest first).
Companion of tools/isa_mix.py; read next to profiles/r03_pmc_sq_counters.txt."""
import collections, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from obca_amd.buildflags import HIPCC      # the product's flags
fn = sys.argv[1] if len(sys.argv) > 1 else "ph_fused2"; which = int(sys.argv[2]) if len(sys.argv) > 2 else 0; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
d = tempfile.mkdtemp()
subprocess.run(HIPCC + [ "-save-temps", "-Wno-error",      # (the preprocessed intermediate loses the macro provenance some warnings are silenced by)
      
                "-gline-tables-only", "-o", "t.so", R + "/obca_amd/csrc/obca_hip.hip"], cwd=d, stderr=subprocess.DEVNULL, check=True)
L = open(os.path.join(d, "obca_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
files = {}
for ln in L:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
    if m: files[int(m.group(1))] = m.group(2)


def cls(op):
    if op.startswith("v_accvgpr"): return "agpr"
    if op.startswith("v_mov_b"): return "mov"
    if re.match(r"v_\w+_f64", op) and not op.startswith("v_cmp"): return "f64"
    if op.startswith(("v_cndmask", "v_cmp")): return "sel"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "s"
    if op.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")): return "mem"
    return None


i = 0
while i < len(L):
    m = re.match(r"^(_Z\w+):", L[i])
    if not m: i += 1; continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    j = i + 1
    while j < len(L) and not L[j].startswith(".Lfunc_end"): j += 1
    if name.endswith(fn):
        loops = collections.OrderedDict(); cur = None; loc = (0, 0)
        for ln in L[i + 1:j]:
            if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", ln) or "Loop Header" in ln or "in Loop" in ln or "Parent Loop" in ln:
                h = re.search(r"(?:in Loop: Header=|Parent Loop )(\w+) Depth=1", ln)
                if h: cur = h.group(1)
                elif re.search(r"Loop Header: Depth=1", ln): cur = "hdr"
                elif re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", ln) and "Loop" not in ln: cur = None
                continue
            m2 = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
            if m2: loc = (int(m2.group(1)), int(m2.group(2))); continue
            t = ln.split()
            if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
                c = cls(t[0])
                if c: loops.setdefault(cur or "outside", collections.defaultdict(collections.Counter))[loc][c] += 1
        keys = [k for k in loops if k not in ("hdr", "outside")] + ["outside"]
        key = keys[which] if which < len(keys) else (keys[-1] if keys else "hdr")
        tab = loops.get(key, {})
        tot = collections.Counter()
        for c in tab.values(): tot.update(c)
        print(name, "loop", key, "instructions by class", dict(tot))
        src = {}
        for (f, l_), c in sorted(tab.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
            fnm = files.get(f, "?")
            if fnm not in src:
                pth = os.path.join(R, "obca_amd", "csrc", fnm); src[fnm] = open(pth).read().splitlines() if os.path.exists(pth) else []
            text = src[fnm][l_ - 1].strip()[:110] if 0 < l_ <= len(src[fnm]) else ""
            print("%4d  %-13s:%-5d %-40s | %s" % (sum(c.values()), fnm, l_, " ".join("%s %d" % kv for kv in c.most_common()), text))
    i = j
