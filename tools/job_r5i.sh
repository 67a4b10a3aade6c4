#!/bin/bash
# round 5, job I: the default bench line, its 20-step form, config 3 and the GPU suite on the final binary -- on a box that first shows it reproduces its own results; a box that
# does not gets the deeper probes instead (hardware unit of every differing instance, pattern probes per CU, poisoned build, round-4 library)
mkdir -p gpurun_out/r5i
O=$PWD/gpurun_out/r5i; C=$PWD/obca_amd/csrc
rocminfo | grep -E "Uuid: +GPU" > $O/uuid.txt; cat $O/uuid.txt
( cd tools/micro && timeout 200 ./cu_consistency 100 ) > $O/cu.txt 2>&1; tail -n 1 $O/cu.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 40 > $O/hw.txt 2>&1; tail -n 1 $O/hw.txt
if ! grep -q "TOTAL differing (instance, run) pairs 0" $O/hw.txt || ! grep -q " 0 deviating" $O/cu.txt; then
  echo "=== this box does not reproduce its own results: deeper probes instead of the bench"
  ( cd tools/micro && timeout 300 ./cu_consistency 400 ) > $O/cu_deep.txt 2>&1; tail -n 14 $O/cu_deep.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 400 python tools/determinism_hw.py 120 > $O/hw_deep.txt 2>&1; grep -E "differing|units of" $O/hw_deep.txt | cut -c1-2500
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_poison.so timeout 300 python tools/determinism.py 12 > $O/poison.txt 2>&1; tail -n 8 $O/poison.txt
  OBCA_HIP_LIBRARY=$C/variants/libobca_hip_r4.so timeout 300 python tools/determinism.py 12 > $O/r4.txt 2>&1; tail -n 8 $O/r4.txt
  ( cd tools/micro && timeout 120 ./lds_atomic_order ) > $O/lds_atomic.txt 2>&1; tail -n 4 $O/lds_atomic.txt
  timeout 300 python tools/determinism_soak.py 60 ipopt > $O/soak.txt 2>&1; tail -n 12 $O/soak.txt | cut -c1-300
  exit 0
fi
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -16 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err
timeout 600 python bench.py --config 3 --no-host-rate --steps 60 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json
for c in ("bench", "bench_driver_line", "bench_cfg3"):
    d = json.loads(open("gpurun_out/r5i/%s.json" % c).read().strip().splitlines()[-1]); k = d["config"]; r = d["roofline"]
    print(c, "value", d["value"], "ms", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "validated", k["converged"], "/", k["instances"], "iters", k["mean_iterations"], "max", k["max_iterations"], "passes", k["mean_passes"], "frac", r["frac"], "regime", r["regime_of_value"]["fp64_frac"], "traffic", r["traffic"], r.get("traffic_over_io_only"), "sync", k["single_batch_sync_solves_per_s"], "bit-identical", k["copies_bit_identical"], "planning", (k.get("planning") or {}).get("seconds"), "e2e", (k.get("planning") or {}).get("end_to_end_solves_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    if k.get("fast_options"): print("   fast", {x: k["fast_options"][x] for x in ("solves_per_s", "mean_iterations", "mean_passes", "solution_differs_from_timed_options")})
    if k.get("other_configs"): print("   other", [(o["config"], o["solves_per_s"], o["validated"], o["fast_options"]["solves_per_s"], (o.get("cpu_baseline") or {}).get("value"), o["batch_made_in_s"]) for o in k["other_configs"]])
    if k.get("host_pointer"): print("   host", k["host_pointer"]["solves_per_s"], k["host_pointer"]["c_call_solves_per_s"], "distinct", (k.get("distinct_batches") or {}).get("solves_per_s"))
PY
