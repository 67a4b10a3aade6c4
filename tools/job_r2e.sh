#!/bin/bash
# round 2, job e: one-wave kernel -- parity suite, PMC traffic of one launch and of the pipelined region, prefetch-depth variants
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2e; R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for V in base fwd8 ric6; do
  OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_$V.so timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$V value',d['value'],'ms',d['ms_per_step'],'kernel_ms',r['kernel_ms'])"
done | tee $O/variants.txt
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc1 -o $C -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc1_$C.err
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc4 -o $C -- python $R/bench.py --steps 16 --warmup 0 --streams 4 --sync-steps 1 --no-cpu-baseline > $O/pmc4_$C.json 2> $O/pmc4_$C.err
done
cd $R; python - <<'PY'
import csv, json
for d in ("pmc1","pmc4"):
    v={}; n={}
    for C in ("FETCH_SIZE","WRITE_SIZE"):
        for r in csv.DictReader(open(f"gpurun_out/r2e/{d}/{C}_counter_collection.csv")):
            if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): v[C]=v.get(C,0)+float(r["Counter_Value"])*1024; n[C]=n.get(C,0)+1
    print(d, "launches", n, "fetch(x2) %.2f GB  write %.2f GB  total %.2f GB" % (2*v["FETCH_SIZE"]/1e9, v["WRITE_SIZE"]/1e9, (2*v["FETCH_SIZE"]+v["WRITE_SIZE"])/1e9))
try:
    j=json.load(open("gpurun_out/r2e/pmc4_FETCH_SIZE.json")); print("pipelined under rocprof: ms_per_step", j["ms_per_step"], "value", j["value"])
except Exception as e: print(e)
PY
