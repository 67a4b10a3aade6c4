#!/bin/bash
# round 4, job A: the instruction-count changes (bounded sin / cos, structural-zero predicates) on a GPU for the first time -- GPU suite, same-box A/B against the round-3 binary,
# the bench line with its new legs (config.ipopt_options, config.other_configs), the IPOPT-configuration census on the full bench batches (kernels AND oracle with the switches on).
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4a; R=$PWD; C=$R/obca_amd/csrc
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 bash tools/ab.sh libobca_hip_base.so libobca_hip.so 2>&1 | tee $O/ab_sync.txt
for L in libobca_hip_base.so libobca_hip.so; do
  OBCA_HIP_LIBRARY=$C/$L timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_pipe_$L.json 2> $O/bench_pipe_$L.err
  python -c "import json;d=json.loads(open('$O/bench_pipe_$L.json').read().strip().splitlines()[-1]);print('$L pipelined', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['converged'])" | tee -a $O/ab_pipelined.txt
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -3 $O/bench.err
timeout 1200 python tools/parity_census.py 2 3 5 --gpu-ipopt-options > $O/census_gpu_ipopt_options.txt 2>&1; grep -v "iteration mismatch" $O/census_gpu_ipopt_options.txt | tail -12
for CF in 3 5; do timeout 600 python bench.py --config $CF --steps 40 --no-cpu-baseline --no-pmc --no-host-rate > $O/bench_cfg$CF.json 2> $O/bench_cfg$CF.err; done
python - <<'PY'
import json
O="gpurun_out/r4a"
for c in ("bench","bench_cfg3","bench_cfg5"):
    try:
        d=json.loads(open(f"{O}/{c}.json").read().strip().splitlines()[-1]); k=d["config"]; r=d.get("roofline") or {}
        print(c,"value",d["value"],"ms",d["ms_per_step"],"kernel_ms",r.get("kernel_ms"),"validated",k.get("converged"),"/",k.get("instances"),"passes",k.get("mean_passes"),"frac",r.get("frac"),"traffic",r.get("traffic"))
        print("   ipopt_options", k.get("ipopt_options")); print("   other_configs", k.get("other_configs")); print("   sync", k.get("single_batch_sync_solves_per_s"), "distinct", (k.get("distinct_batches") or {}).get("solves_per_s"), "host", (k.get("host_pointer") or {}).get("c_call_solves_per_s"), "cpu", d.get("cpu_baseline"))
    except Exception as e: print(c, "ERR", e)
PY
