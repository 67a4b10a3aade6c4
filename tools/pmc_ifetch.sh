#!/bin/bash
# instruction-fetch counters of the parking IPM kernel (one rocprofv3 pass per group): the phases are straight-line code of 10-76 KB each, ~105 KB per pass, against an
# instruction cache of 64 KB shared by two CUs -- does the front end stall the wavefronts?
R=$(pwd); export TMPDIR=/tmp; cd /tmp
B=${1:-1024}
rm -rf $R/gpurun_out/pmcif; mkdir -p $R/gpurun_out/pmcif
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcif -o g$i -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs --batch $B > /dev/null 2> $R/gpurun_out/pmcif/g$i.err || tail -3 $R/gpurun_out/pmcif/g$i.err
done
cd $R; python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmcif/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("%-32s launches %d  mean %.6g  max %.6g" % (k, len(v), sum(v) / len(v), max(v)))
PY
