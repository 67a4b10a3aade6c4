#!/bin/bash
# round 4, job E: what saturates under load?  Memory-path counters (TA / TCP / TCC) and SQ wait counters of the parking kernel with the machine full (B = 8 192: 4 instances per CU
# for the whole launch -- rocprofv3 serialises kernels, so the pipelined regime itself cannot be counted)
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4e; R=$PWD
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "\b\(TA\|TCP\|TCC\|TD\|SQ\|GRBM\)_[A-Za-z0-9_]*" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
i=0
for C in "TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
         "TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_EA0_WRREQ_64B_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" \
         "GRBM_GUI_ACTIVE GRBM_COUNT MemUnitBusy MemUnitStalled" "WriteUnitStalled L2CacheHit VALUBusy MeanOccupancyPerCU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/bench.py --batch 8192 --steps 2 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > /dev/null 2> $O/g$i.err || (echo "group $i failed: $C"; tail -2 $O/g$i.err)
done
cd $R; python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r4e/g*/**/*counter_collection.csv", recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("obca_parking_ipm_kernel"): acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-40s launches %d  mean %.6g  max %.6g" % (k, len(v), sum(v) / len(v), max(v)))
PY
