#!/bin/bash
# round 5, job F: the GPU suite in its final order (small-batch parity of both kernels, full-machine parity, determinism, plumbing) with the self-test line, then the bench line
mkdir -p gpurun_out/r5f
O=$PWD/gpurun_out/r5f
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -n 16 $O/pytest_gpu.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -n 3 $O/bench.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f/bench.json").read().strip().splitlines()[-1])
k = d["config"]; r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "| fast", k["fast_options"]["solves_per_s"], "differs", k["fast_options"]["solution_differs_from_timed_options"], "copies_bit_identical", k["copies_bit_identical"])
print("roofline", {x: r[x] for x in ("bound", "achieved", "frac", "traffic", "kernel_ms", "traffic_over_io_only") if x in r})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for o in k["other_configs"]:
    print(o["config"], o["solves_per_s"], o["validated"], "fast", o["fast_options"]["solves_per_s"], "cpu", o["cpu_baseline"]["value"] if o["cpu_baseline"] else None, o["batch_made_in_s"])
PY
