#!/bin/bash
# round 2, job i: full GPU suite + quadcopter profiles after the MFMA sweep / one-wavefront kernel
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2i; R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py --config 4 --no-cpu-baseline --steps 24 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cut -c1-300 $O/bench_cfg4.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $C -- python $R/bench.py --config 4 --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_cfg4_$C.err
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python $R/bench.py --config 4 --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mfma.err
cd $R; tail -3 $O/pmc_mfma.err
python - <<'PY'
import csv, glob
O="gpurun_out/r2i"
for d in ("pmc_cfg4","pmc_mfma"):
    v={}
    for f in glob.glob(f"{O}/{d}/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]; v.setdefault(k,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,c in v.items(): print(d, k, {C:(len(x), sum(x)) for C,x in c.items()})
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv"): print(open(f).read()[:700])
PY
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 64 > $O/quad_phase_B64.txt; OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 1024 > $O/quad_phase_B1024.txt
