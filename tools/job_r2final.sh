#!/bin/bash
# round 2, final job: GPU suite, smoke, the bench lines of all configs, PCIe-inclusive rates, rocprofv3 kernel stats of synchronous steps, PMC traffic of the
# parking and quadcopter kernels, MFMA instruction counters, per-phase clocks, streams sweep, the end-to-end example
mkdir -p gpurun_out/r2w
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2w; R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sync -o t -- python $R/bench.py --steps 8 --warmup 2 --streams 1 --sync-steps 4 --no-cpu-baseline > $O/bench_sync_under_rocprof.json 2> $O/stats_sync.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cfg2 -o $C -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_cfg2_$C.err
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $C -- python $R/bench.py --config 4 --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_cfg4_$C.err
done
timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python $R/bench.py --config 4 --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mfma.err
cd $R
# the bench lines below quote roofline.traffic from the PMC passes of THIS build: same files as committed under profiles/
cp $O/pmc_cfg2/FETCH_SIZE_counter_collection.csv profiles/r02_pmc_fetch_size.csv; cp $O/pmc_cfg2/WRITE_SIZE_counter_collection.csv profiles/r02_pmc_write_size.csv
cp $O/pmc_cfg4/FETCH_SIZE_counter_collection.csv profiles/r02_pmc_quad_fetch_size.csv; cp $O/pmc_cfg4/WRITE_SIZE_counter_collection.csv profiles/r02_pmc_quad_write_size.csv
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-250 $O/bench.json
timeout 900 python bench.py --config 3 --no-cpu-baseline --steps 56 > $O/bench_cfg3.json 2>/dev/null
timeout 900 python bench.py --config 4 --no-cpu-baseline --steps 36 > $O/bench_cfg4.json 2>/dev/null
timeout 900 python bench.py --config 5 --no-cpu-baseline --steps 28 > $O/bench_cfg5.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --warm-start hybrid --steps 160 > $O/bench_cfg2_hybrid.json 2>/dev/null
for S_ in 2 6 8; do timeout 300 python bench.py --no-cpu-baseline --steps 120 --streams $S_ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams',d['config']['streams'],'value',d['value'],'ms',d['ms_per_step'])"; done | tee $O/streams_sweep.txt
timeout 600 python tools/pcie_rate.py 4096 16384 > $O/pcie_rate.json 2>/dev/null; cat $O/pcie_rate.json
timeout 300 python examples/main_parking.py 2>&1 | tail -6 | tee $O/main_parking.txt
cd $R
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/phase_profile.py 64 > $O/phase_B64.txt; OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/phase_profile.py 1024 > $O/phase_B1024.txt
OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 64 > $O/quad_phase_B64.txt; OBCA_HIP_LIBRARY=$R/obca_amd/csrc/libobca_hip_prof.so python tools/quad_gpu.py 1024 > $O/quad_phase_B1024.txt
python - <<'PY'
import csv, glob, json
O="gpurun_out/r2w"
for c in ("bench","bench_cfg3","bench_cfg4","bench_cfg5","bench_cfg2_hybrid"):
    try:
        d=json.load(open(f"{O}/{c}.json")); k=d["config"]; r=d["roofline"]
        print(c,"value",d["value"],"ms",d["ms_per_step"],"kernel_ms",r["kernel_ms"],"validated",k["converged"],"/",k["instances"],"iters",k["mean_iterations"],"passes",k["mean_passes"],"frac",r["frac"], "hbm_pipelined", r.get("hbm_measured_gbs_pipelined"))
    except Exception as e: print(c, "ERR", e)
for d in ("pmc_cfg2","pmc_cfg4","pmc_mfma"):
    v={}
    for f in glob.glob(f"{O}/{d}/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            if "ipm" in k: v.setdefault(k,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,c in v.items(): print(d, k, {C:(len(x), sum(x)/len(x)) for C,x in c.items()})
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv"): print(open(f).read()[:500])
PY
