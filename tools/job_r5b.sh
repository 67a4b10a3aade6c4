#!/bin/bash
# round 5, job B: the GPU suite in its new order with the determinism tests, the poisoned build under the reference's IPOPT configuration, the restructured bench line
mkdir -p gpurun_out/r5b
O=$PWD/gpurun_out/r5b; C=$PWD/obca_amd/csrc
timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; tail -n 25 $O/pytest_gpu.log
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python tools/determinism_ragged.py 10 40 150 reference > $O/ragged_poison_reference_opts.txt 2>&1; tail -n 3 $O/ragged_poison_reference_opts.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python -m pytest tests/test_gpu_quad_parity.py -m gpu -q -x > $O/pytest_quad_poison.log 2>&1; tail -n 3 $O/pytest_quad_poison.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -n 3 $O/bench.time; tail -c 600 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
k = d["config"]; r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "options", k["options"][:40], "| fast", k["fast_options"]["solves_per_s"], "differs", k["fast_options"]["solution_differs_from_timed_options"])
print("roofline", {x: r[x] for x in ("bound", "achieved", "frac", "traffic", "kernel_ms", "hbm_io_only_frac", "implementation_hbm_frac", "traffic_over_io_only") if x in r}, r["regime_of_value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for o in k["other_configs"]:
    print(o["config"], o["solves_per_s"], o["validated"], "fast", o["fast_options"]["solves_per_s"], "cpu", o["cpu_baseline"]["value"] if o["cpu_baseline"] else None, o["batch_made_in_s"])
print("sync", k["single_batch_sync_solves_per_s"], "host", k["host_pointer"]["solves_per_s"] if k["host_pointer"] else None, "distinct", k["distinct_batches"]["solves_per_s"] if k["distinct_batches"] else None)
PY
