"""Static instruction mix of functions of the gfx950 code object, per top-level loop: python tools/isa_mix.py ph_fused2 ph_direction2 ph_riccati [-- extra hipcc flags]
(hipcc -save-temps assembly; a loop body's count x its trip count is the dynamic count: the obstacle loop of a phase runs ceil((N + 1) nOb / 64) times, the stage loop
ceil((N + 1) / 64) times).  Read next to the SQ counters of tools/pmc_sq.sh (profiles/r03_pmc_sq_counters.txt): which share of the issued VALU instructions is fp64 arithmetic."""
import collections, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from obca_amd.buildflags import HIPCC      # the product's flags
args = sys.argv[1:]; extra = []
if "--" in args: i = args.index("--"); extra = args[i + 1:]; args = args[:i]
want = args or ["ph_fused2", "ph_direction2", "ph_riccati"]
d = tempfile.mkdtemp()
subprocess.run(HIPCC + [ "-save-temps", "-Wno-error",      # (the preprocessed intermediate loses the macro provenance some warnings are silenced by)
       "-o", "t.so",
                R + "/obca_amd/csrc/obca_hip.hip"] + extra, cwd=d, stderr=subprocess.DEVNULL, check=True)
lines = open(os.path.join(d, "obca_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()


def cls(op):
    if op.startswith("v_accvgpr"): return "AGPR moves"
    if op.startswith("v_mov_b"): return "v_mov"
    if op.startswith("v_mfma"): return "mfma"
    if re.match(r"v_\w+_f64", op) and not op.startswith("v_cmp"): return "fp64 arithmetic"
    if op.startswith(("v_cndmask", "v_cmp")): return "compare / select"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")) or "dpp" in op: return "lane exchange"
    if op.startswith("v_"): return "other VALU (addresses, integers)"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "memory"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_"): return "scalar"
    return None


VALU = ("AGPR moves", "v_mov", "fp64 arithmetic", "compare / select", "lane exchange", "other VALU (addresses, integers)", "mfma")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not m: i += 1; continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"): j += 1
    if any(w in name for w in want):
        loops = collections.OrderedDict(); cur = "outside loops"; k = i + 1
        while k < j:
            ln = lines[k]
            if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", ln):       # a block: its loop tag is in the comment of this and the following comment lines
                blk = ln.split(":")[0]; txt = ln; q = k + 1
                while q < j and lines[q].lstrip().startswith(";") and not re.match(r"^; %bb", lines[q]): txt += lines[q]; q += 1
                h = re.search(r"(?:in Loop: Header=|Parent Loop )(\w+) Depth=1", txt)
                cur = ("loop " + h.group(1)) if h else (("loop " + blk.lstrip(".L")) if re.search(r"Loop Header: Depth=1", txt) else "outside loops")
            else:
                t = ln.split()
                if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
                    c = cls(t[0])
                    if c: loops.setdefault(cur, collections.Counter())[c] += 1
            k += 1
        print(name)
        for lp, c in loops.items():
            tot = sum(c.values()); valu = sum(c[x] for x in VALU)
            if tot < 40: continue
            print("  %-22s %5d instructions, %5d VALU: " % (lp, tot, valu) + ", ".join("%s %d (%.0f %%)" % (x, c[x], 100.0 * c[x] / max(valu, 1)) for x in VALU if c[x]) +
                  " | " + ", ".join("%s %d" % (x, c[x]) for x in ("LDS", "memory", "scalar", "s_waitcnt") if c[x]))
    i = j
