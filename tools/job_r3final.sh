#!/bin/bash
# round 3, final job (run as `gpurun -- bash tools/job_r3final.sh`): GPU suite, smoke, the default bench line (live PMC traffic, CPU baseline, host-pointer rate), the bench lines of
# configs 3 / 4 / 5, the single-process route, 2-rank gloo lines of configs 2-5, rocprofv3 kernel stats of synchronous steps, PMC passes of both IPM kernels, per-phase clocks.
# Everything lands in gpurun_out/r3final/; the files DESIGN.md cites are copied to profiles/r03_* afterwards (tools/README.md).
mkdir -p gpurun_out/r3final
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3final; R=$PWD; C=$R/obca_amd/csrc
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sync -o t -- python $R/bench.py --steps 8 --warmup 2 --streams 1 --sync-steps 4 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct > $O/bench_sync_under_rocprof.json 2> $O/stats_sync.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
for K in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg2 -o $K -- python $R/bench.py --pmc-child > /dev/null 2> $O/pmc_cfg2_$K.err
  timeout 600 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $K -- python $R/bench.py --config 4 --pmc-child > /dev/null 2> $O/pmc_cfg4_$K.err
done
cd $R
for CF in 3 4 5; do timeout 900 python bench.py --config $CF --no-cpu-baseline --no-host-rate --steps 60 > $O/bench_cfg$CF.json 2> $O/bench_cfg$CF.err; done
timeout 600 python bench.py --single-process --steps 60 > $O/bench_single_process.json 2> $O/sp.err
for CF in 2 3 4 5; do
  t0=$(date +%s.%N)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+CF)) bench.py --gpus 2 --backend gloo --config $CF --steps 24 --warmup 4 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_2rank_gloo_cfg$CF.json 2> $O/bench_2rank_gloo_cfg$CF.err
  t1=$(date +%s.%N); echo "2 ranks gloo config $CF: whole run $(python -c "print('%.1f' % ($t1-$t0))") s" | tee -a $O/two_rank_wall.txt
done
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/quad_gpu.py $B > $O/quad_phase_B$B.txt 2>&1; done
python - <<'PY'
import csv, glob, json
O="gpurun_out/r3final"
for c in ("bench","bench_cfg3","bench_cfg4","bench_cfg5","bench_single_process","bench_2rank_gloo_cfg2","bench_2rank_gloo_cfg3","bench_2rank_gloo_cfg4","bench_2rank_gloo_cfg5"):
    try:
        d=json.loads(open(f"{O}/{c}.json").read().strip().splitlines()[-1]); k=d["config"]; r=d.get("roofline") or {}
        print(c,"value",d["value"],"ms",d["ms_per_step"],"kernel_ms",r.get("kernel_ms"),"validated",k.get("converged"),"/",k.get("instances", k.get("instances_per_step")),"passes",k.get("mean_passes"),"bound",r.get("bound"),r.get("frac"),"traffic",r.get("traffic"),r.get("traffic_over_algorithmic"),"sync",k.get("single_batch_sync_solves_per_s"),"host",k.get("host_pointer"),"planning",k.get("planning"), "cpu", d.get("cpu_baseline"))
    except Exception as e: print(c, "ERR", e)
for d in ("pmc_cfg2","pmc_cfg4"):
    v={}
    for f in glob.glob(f"{O}/{d}/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            if "ipm" in k: v.setdefault(k,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,c in v.items(): print(d, k, {C_:(len(x), sum(x)/len(x)) for C_,x in c.items()})
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv"): print(open(f).read()[:400])
PY
