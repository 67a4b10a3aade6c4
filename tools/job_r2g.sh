#!/bin/bash
# round 2, job g: the profiles of the round -- GPU suite, default bench line, rocprofv3 kernel stats of synchronous steps, PMC traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) for the parking, quadcopter and DualMultWS kernels, counter calibration on 8-byte gathers / Infinity-Cache re-reads, bench lines of configs 3 / 4 / 5
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2g; R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sync -o t -- python $R/bench.py --steps 8 --warmup 2 --streams 1 --sync-steps 4 --no-cpu-baseline > $O/bench_sync_under_rocprof.json 2> $O/stats_sync.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cfg2 -o $C -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_cfg2_$C.err
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $C -- python $R/bench.py --config 4 --steps 1 --warmup 0 --streams 1 --sync-steps 1 --no-cpu-baseline > /dev/null 2> $O/pmc_cfg4_$C.err
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib -o $C -- $R/tools/micro/fetch_calib > /dev/null 2> $O/pmc_calib_$C.err
done
cd $R
timeout 900 python bench.py --config 3 --no-cpu-baseline --steps 40 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cat $O/bench_cfg3.json | cut -c1-400
timeout 900 python bench.py --config 4 --no-cpu-baseline --steps 12 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cat $O/bench_cfg4.json | cut -c1-400
timeout 900 python bench.py --config 5 --no-cpu-baseline --steps 16 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cat $O/bench_cfg5.json | cut -c1-400
python - <<'PY'
import csv, glob
O="gpurun_out/r2g"
for d in ("pmc_cfg2","pmc_cfg4","pmc_calib"):
    v={}
    for C in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob(f"{O}/{d}/{C}_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"].split("(")[0][:40]; v.setdefault(k,{}).setdefault(C,[]).append(float(r["Counter_Value"])*1024)
    for k,c in v.items(): print(d, k, {C:(len(x), round(sum(x)/1e9,3)) for C,x in c.items()})
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv"): print(f); print(open(f).read()[:1500])
PY
