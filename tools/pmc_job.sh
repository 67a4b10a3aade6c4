#!/bin/bash
# PMC traffic of one bench launch (separate FETCH_SIZE / WRITE_SIZE passes), printed as bytes per launch of the IPM kernel
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pmcq; mkdir -p $R/gpurun_out/pmcq
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcq -o $C -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmcq/$C.err
done
cd $R; python - <<'PY'
import csv
v={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    for r in csv.DictReader(open(f"gpurun_out/pmcq/{C}_counter_collection.csv")):
        if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): v[C]=float(r["Counter_Value"])*1024
print("fetch(x2) %.2f GB  write %.2f GB  total %.2f GB" % (2*v["FETCH_SIZE"]/1e9, v["WRITE_SIZE"]/1e9, (2*v["FETCH_SIZE"]+v["WRITE_SIZE"])/1e9))
PY
