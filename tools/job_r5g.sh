#!/bin/bash
# round 5, job G: after the planner's default went to the tail-neutral setting (Reeds-Shepp heuristic, weight 1.5): GPU suite, default bench line, config-3 line
mkdir -p gpurun_out/r5g
O=$PWD/gpurun_out/r5g
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -16 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err
timeout 600 python bench.py --config 3 --no-host-rate --steps 60 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
for c in ("bench", "bench_driver_line", "bench_cfg3"):
    d = json.loads(open("gpurun_out/r5g/%s.json" % c).read().strip().splitlines()[-1]); k = d["config"]; r = d["roofline"]
    print(c, "value", d["value"], "ms", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "validated", k["converged"], "/", k["instances"], "iters", k["mean_iterations"], "max", k["max_iterations"], "passes", k["mean_passes"], "frac", r["frac"], "sync", k["single_batch_sync_solves_per_s"], "planning", (k.get("planning") or {}).get("seconds"), "e2e", (k.get("planning") or {}).get("end_to_end_solves_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    if k.get("fast_options"): print("   fast", {x: k["fast_options"][x] for x in ("solves_per_s", "mean_iterations", "mean_passes", "solution_differs_from_timed_options")})
    if k.get("other_configs"): print("   other", [(o["config"], o["solves_per_s"], o["validated"], o["fast_options"]["solves_per_s"], (o.get("cpu_baseline") or {}).get("value"), o["batch_made_in_s"]) for o in k["other_configs"]])
PY
