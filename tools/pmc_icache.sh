#!/bin/bash
# instruction-cache / scalar-cache / LDS counters of the IPM kernel for one bench launch (one rocprofv3 pass per group): what do the four wavefronts of a CU contend for?
R=$(pwd); export TMPDIR=/tmp; cd /tmp
B=${1:-1024}
rm -rf $R/gpurun_out/pmcic; mkdir -p $R/gpurun_out/pmcic
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|DCACHE|LDS|INST_LEVEL|WAIT_INST|INSTS_SMEM|TC_REQ|TC_INST)[A-Z_0-9]*)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmcic/available.txt
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_BUSY_CYCLES" "SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcic -o g$i -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs --batch $B > /dev/null 2> $R/gpurun_out/pmcic/g$i.err || echo "group $i failed: $(tail -2 $R/gpurun_out/pmcic/g$i.err | cut -c1-200)"
done
cd $R; python - <<'PY'
import csv, glob
seen = set()
for f in sorted(glob.glob("gpurun_out/pmcic/**/g*_counter_collection.csv", recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-32s %16.0f   (launches %d)" % (k, v[-1], len(v)))
PY
