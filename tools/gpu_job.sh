#!/bin/bash
# ONE parameterised GPU job (replaces the per-experiment tools/job_r3*.sh .. job_r5*.sh of rounds 3-5; those are in the git history, what they measured is in docs/HISTORY.md):
#     gpurun --timeout S -- bash tools/gpu_job.sh TAG step [step ...]
# writes everything under gpurun_out/TAG/ (merged back by gpurun; the files to be judged are then copied to profiles/ by hand).  Steps:
#   stamp        git revision + md5 of the library + kernel symbols
#   tests        pytest -m gpu (the whole GPU suite, parity first -- tests/conftest.py orders it), the census files of the tolerant tests
#   smoke        __graft_entry__.smoke()
#   bench        the default bench line (value, roofline with live PMC traffic, cpu_baseline, other configs)
#   bench20      the driver's invocation: --gpus 1 --steps 20 --warmup 5
#   quick        a short default-config line without the side legs (A/B of kernel changes): value, kernel_ms, single-batch rate
#   cfg3 cfg4 cfg5   bench.py --config C
#   stats        rocprofv3 --kernel-trace --stats of synchronous steps, configs 2 and 4 (kernel_stats.csv)
#   pmc          FETCH_SIZE / WRITE_SIZE passes of both IPM kernels (separate passes, as MI355X_MICROARCH.md prescribes)
#   sq           SQ counter groups of the parking kernel (tools/pmc_sq.sh) ; mfma: the quadcopter kernel's MFMA counters
#   phase        per-phase clocks of the parking kernel, -DOBCA_PROFILE build, B = 64 and 1024, both option sets ; phase5 / quadphase likewise for config 5 / the quadcopter kernel
#   census       tools/options_census.py 2 3 5
#   sched        scheduling experiments: --streams 4 / 6 / 8 / 12; OBCA_SLICE_ALWAYS=1 with slices of 4 / 8 passes
#   micro        the micro-benchmarks behind DESIGN.md section 5 (fp64 dependent latency, the sweep's phase pattern with 1-3 chains, the parking stage on MFMA tiles)
#   queues       GPU_MAX_HW_QUEUES x --streams on the pipelined line ; slots: residency of the SIMDs (tools/load_profile.py, slot_timeline.py), phase clocks against residency ; icache: tools/pmc_icache.sh
#   gloo2        bench.py --gpus 2 --backend gloo (two ranks on the one GPU)
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$TAG; C=$R/obca_amd/csrc; mkdir -p $O
LEAN="--no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs"
summ() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["config"]; r = d.get("roofline") or {}
        print(f.split("/")[-1], "value", d["value"], "ms", d["ms_per_step"], "kernel_ms", r.get("kernel_ms"), "validated", k.get("converged"), "/", k.get("instances"), "iters", k.get("mean_iterations"), "passes", k.get("mean_passes"),
              "frac", r.get("frac"), "regime", (r.get("regime_of_value") or {}).get("frac"), "traffic/io", r.get("traffic_over_io_only"), "single", k.get("single_batch_sync_solves_per_s"))
        if k.get("fast_options"): print("   fast", {x: k["fast_options"].get(x) for x in ("solves_per_s", "mean_iterations", "mean_passes")})
        if k.get("other_configs"): print("   other", [(o["config"], o["solves_per_s"], o["validated"], o["fast_options"]["solves_per_s"], (o.get("cpu_baseline") or {}).get("value")) for o in k["other_configs"]])
        if k.get("host_pointer"): print("   host", k["host_pointer"].get("c_call_solves_per_s"), "distinct", (k.get("distinct_batches") or {}).get("solves_per_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
}
for STEP in "$@"; do
  echo "=== $STEP"
  case $STEP in
    stamp) { echo "revision: $(cat $R/.revision 2>/dev/null)"; echo "library: $(md5sum $C/libobca_hip.so | cut -c1-12)"; rocminfo | grep -E "Uuid: +GPU"; /opt/rocm/lib/llvm/bin/llvm-nm -C --defined-only $C/libobca_hip.so 2>/dev/null | grep -i "ipm_kernel" | head -4; } > $O/STAMP.txt; cat $O/STAMP.txt ;;
    tests) timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log | cut -c1-400; cp gpurun_out/parity_census_*.txt $O/ 2>/dev/null ;;
    smoke) timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log ;;
    bench) timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; summ $O/bench.json ;;
    bench20) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; summ $O/bench_driver_line.json ;;
    quick) timeout 300 python bench.py --steps 40 --warmup 8 $LEAN > $O/bench_quick.json 2> $O/bench_quick.err; summ $O/bench_quick.json ;;
    sched) # scheduling experiments on the pipelined default line: more streams in flight; the two-launch schedule forced for resident batches (OBCA_SLICE_ALWAYS)
      for S in 4 6 8 12; do timeout 300 python bench.py --steps 60 --warmup 12 --streams $S $LEAN > $O/bench_streams$S.json 2> $O/bench_streams$S.err; echo "streams $S"; summ $O/bench_streams$S.json; done
      for S in 4 8; do for P in 4 8; do OBCA_SLICE_ALWAYS=1 OBCA_SLICE_PASSES=$P timeout 300 python bench.py --steps 60 --warmup 12 --streams $S $LEAN > $O/bench_slice${P}_streams$S.json 2> $O/bench_slice${P}_streams$S.err; echo "sliced $P passes, streams $S"; summ $O/bench_slice${P}_streams$S.json; done; done ;;
    cfg3|cfg4|cfg5) CF=${STEP#cfg}; timeout 900 python bench.py --config $CF --no-host-rate --steps 60 > $O/bench_cfg$CF.json 2> $O/bench_cfg$CF.err; summ $O/bench_cfg$CF.json ;;
    stats) cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sync -o t -- python $R/bench.py --steps 8 --warmup 2 --streams 1 --sync-steps 4 $LEAN > $O/bench_sync_under_rocprof.json 2> $O/stats_sync.err
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline --no-pmc --no-host-rate --no-ipopt-leg > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
      cd $R; for f in $(find $O/stats_sync $O/stats_quad -name "*kernel_stats.csv"); do echo $f; head -6 $f | cut -c1-200; done ;;
    pmc) cd /tmp
      for K in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg2 -o $K -- python $R/bench.py --pmc-child > /dev/null 2> $O/pmc_cfg2_$K.err
        timeout 300 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $K -- python $R/bench.py --config 4 --pmc-child > /dev/null 2> $O/pmc_cfg4_$K.err
      done
      cd $R; python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for d in ("pmc_cfg2", "pmc_cfg4"):
    v = {}
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "ipm" in k: v.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, c in v.items(): print(d, k, {C_: (len(x), sum(x) / len(x)) for C_, x in c.items()})
PY
      ;;
    queues) # hardware queues x streams of the pipelined default line (GPU_MAX_HW_QUEUES: the runtime's default of 4 serialises streams that share a queue)
      for QS in "4 4" "4 8" "4 16" "8 8" "16 8" "16 16" "16 32" "24 24" "32 32"; do set -- $QS; echo "queues $1 streams $2"; GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --steps 192 --warmup 32 --streams $2 $LEAN > $O/bench_q$1_s$2.json 2> $O/bench_q$1_s$2.err; summ $O/bench_q$1_s$2.json; done ;;
    slots) # who holds the 1 024 SIMDs: residency / shader clock with 4 and 16 batches in flight, slot timeline of 4, 8, 16 launches queued at once, per-phase clocks against the wavefronts resident
      for QS in "4 4 48" "16 16 96"; do set -- $QS; GPU_MAX_HW_QUEUES=$1 OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/load_profile.py $2 $3 > $O/load_profile_q$1_s$2.txt 2>&1; head -22 $O/load_profile_q$1_s$2.txt; done
      for K in 4 8 16; do GPU_MAX_HW_QUEUES=16 OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/slot_timeline.py $K > $O/slot_timeline_$K.txt 2>&1; head -8 $O/slot_timeline_$K.txt; done
      for B in 64 256 512 1024; do echo "== $B instances (of 1 024 SIMDs) in one launch"; OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B ipopt 2>&1 | grep -v "slowest instance\|options:\|ric_p"; done > $O/phase_clocks_against_residency.txt 2>&1; cat $O/phase_clocks_against_residency.txt ;;
    micro) for M in fp64_dependent_latency riccati_two_chains parking_stage_mfma; do (cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o /tmp/$M $M.hip 2>/dev/null) && timeout 120 /tmp/$M > $O/micro_$M.txt 2>&1; head -30 $O/micro_$M.txt; done ;;
    icache) timeout 900 bash tools/pmc_icache.sh 1024 > $O/pmc_icache_B1024.txt 2>&1; timeout 900 bash tools/pmc_icache.sh 256 > $O/pmc_icache_B256.txt 2>&1; tail -26 $O/pmc_icache_B1024.txt; tail -26 $O/pmc_icache_B256.txt ;;
    sq) timeout 600 bash tools/pmc_sq.sh 1024 > $O/pmc_sq.txt 2>&1; tail -24 $O/pmc_sq.txt ;;
    mfma) cd /tmp; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_quad_mfma -o mfma -- python $R/bench.py --config 4 --pmc-child > /dev/null 2> $O/pmc_quad_mfma.err; cd $R
      python - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/pmc_quad_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "quad_ipm" in r["Kernel_Name"]: print("%-32s %16.0f" % (r["Counter_Name"], float(r["Counter_Value"])))
PY
      ;;
    phase) for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B ipopt > $O/phase_B${B}_reference_options.txt 2>&1; done; cat $O/phase_B1024_reference_options.txt; head -3 $O/phase_B1024.txt; head -3 $O/phase_B64_reference_options.txt ;;
    phaseab) for V in ipopt ipopt-norestore; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py 1024 $V > $O/phase_B1024_$V.txt 2>&1; head -5 $O/phase_B1024_$V.txt | cut -c1-600; done ;;
    phase5) OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 300 python tools/phase_profile5.py > $O/phase_config5.txt 2>&1; cat $O/phase_config5.txt ;;
    census) timeout 900 python tools/options_census.py 2 3 5 > $O/options_census.txt 2>&1; cat $O/options_census.txt ;;
    gloo2) timeout 300 python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 $LEAN > $O/bench_2rank_gloo_selflaunch.json 2> $O/bench_2rank_gloo_selflaunch.err; summ $O/bench_2rank_gloo_selflaunch.json ;;
    *) echo "unknown step $STEP" ;;
  esac
done
