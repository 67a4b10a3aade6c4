#!/bin/bash
# SQ instruction / cycle counters of the IPM kernel for one bench launch (one rocprofv3 pass per group): issue-bound or latency-bound?
R=$(pwd); export TMPDIR=/tmp; cd /tmp
B=${1:-1024}
rm -rf $R/gpurun_out/pmcsq; mkdir -p $R/gpurun_out/pmcsq
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq -o g$i -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs --batch $B > /dev/null 2> $R/gpurun_out/pmcsq/g$i.err || tail -3 $R/gpurun_out/pmcsq/g$i.err
done
cd $R; python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmcsq/g*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): print("%-28s %16.0f" % (r["Counter_Name"], float(r["Counter_Value"])))
PY
