#!/bin/bash
# round 3, after the last quadcopter kernel change (23 MFMAs per stage): GPU suite, smoke, config-4 bench line (live PMC traffic), rocprofv3 kernel stats of synchronous quadcopter
# steps, per-phase clocks, the 2-rank gloo line.  Lands in gpurun_out/r3quad/; copied to profiles/r03_* afterwards (tools/README.md).
mkdir -p gpurun_out/r3quad
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3quad; R=$PWD; C=$R/obca_amd/csrc
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
timeout 400 python bench.py --config 4 --no-cpu-baseline --no-host-rate --steps 60 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
cd $R
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29604 bench.py --gpus 2 --backend gloo --config 4 --steps 24 --warmup 4 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_2rank_gloo_cfg4.json 2> $O/bench_2rank_gloo_cfg4.err
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/quad_gpu.py $B > $O/quad_phase_B$B.txt 2>&1; done
python - <<'PY'
import json, glob
O="gpurun_out/r3quad"
for c in ("bench_cfg4","bench_2rank_gloo_cfg4"):
    try:
        d=json.loads(open(f"{O}/{c}.json").read().strip().splitlines()[-1]); k=d["config"]; r=d.get("roofline") or {}
        print(c,"value",d["value"],"ms",d["ms_per_step"],"kernel_ms",r.get("kernel_ms"),"validated",k.get("converged"),"passes",k.get("mean_passes"),"bound",r.get("bound"),r.get("frac"),"traffic",r.get("traffic"))
    except Exception as e: print(c, "ERR", e)
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv"): print(open(f).read()[:400])
PY
head -3 $O/quad_phase_B1024.txt
