#!/bin/bash
# round 5, final job (run as `gpurun -- bash tools/job_r5final.sh`): GPU suite, smoke, the default bench line (value with the reference's IPOPT configuration, live PMC traffic, CPU baselines,
# host-pointer rate, throughput-options leg, configs 3 / 4 / 5), the bench lines of configs 3 / 4 / 5, rocprofv3 kernel stats of synchronous steps (configs 2 and 4), FETCH / WRITE PMC passes of both IPM kernels, the quadcopter
# kernel's MFMA counters, SQ counters, per-phase clocks, a 2-rank gloo line, the self-launching `bench.py --gpus 2`.  Everything lands in gpurun_out/r5final/ with a stamp of the git
# revision and the kernel signatures of the library that produced it; the files DESIGN.md cites are copied to profiles/r04_* afterwards.
mkdir -p gpurun_out/r5final
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5final; R=$PWD; C=$R/obca_amd/csrc
{ echo "revision: $(cat $R/.revision 2>/dev/null)"; echo "library: $(md5sum $C/libobca_hip.so | cut -c1-12)"; /opt/rocm/lib/llvm/bin/llvm-nm -C --defined-only $C/libobca_hip.so 2>/dev/null | grep -i "ipm_kernel" | head -4; } > $O/STAMP.txt
python - >> $O/STAMP.txt <<'PY'
import subprocess, re, glob, os
lib = os.path.join(os.environ.get("PWD", "."), "obca_amd/csrc/libobca_hip.so")
try:
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", "--input=" + lib], capture_output=True, text=True).stdout
    print("bundles:", out.strip().replace("\n", " "))
except Exception as e: print("bundler:", e)
s = open(lib, "rb").read()
for m in sorted(set(re.findall(rb"_Z\d+obca_(?:parking|quad)_ipm_kernel\w+", s))): print("kernel symbol:", m.decode(), "=", subprocess.run(["c++filt", m.decode()], capture_output=True, text=True).stdout.strip())
PY
cat $O/STAMP.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -16 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; cut -c1-160 $O/bench_driver_line.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sync -o t -- python $R/bench.py --steps 8 --warmup 2 --streams 1 --sync-steps 4 --no-cpu-baseline --no-pmc --no-host-rate --no-distinct --no-ipopt-leg --no-other-configs > $O/bench_sync_under_rocprof.json 2> $O/stats_sync.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_quad -o t -- python $R/bench.py --config 4 --steps 4 --warmup 1 --streams 1 --sync-steps 2 --no-cpu-baseline --no-pmc --no-host-rate --no-ipopt-leg > $O/bench_quad_sync_under_rocprof.json 2> $O/stats_quad.err
for K in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg2 -o $K -- python $R/bench.py --pmc-child > /dev/null 2> $O/pmc_cfg2_$K.err
  timeout 300 rocprofv3 --pmc $K --kernel-trace --output-format csv -d $O/pmc_cfg4 -o $K -- python $R/bench.py --config 4 --pmc-child > /dev/null 2> $O/pmc_cfg4_$K.err
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_quad_mfma -o mfma -- python $R/bench.py --config 4 --pmc-child > /dev/null 2> $O/pmc_quad_mfma.err
cd $R
timeout 600 bash tools/pmc_sq.sh 1024 > $O/pmc_sq.txt 2>&1; tail -22 $O/pmc_sq.txt
timeout 300 bash tools/pmc_ifetch.sh 1024 > $O/pmc_ifetch.txt 2>&1; tail -16 $O/pmc_ifetch.txt
timeout 300 python tools/quad_soc_ab.py 1024 2>&1 | tail -1 > $O/quad_ipopt_switches.txt; timeout 300 python tools/quad_soc_ab.py 4096 2>&1 | tail -1 >> $O/quad_ipopt_switches.txt; cat $O/quad_ipopt_switches.txt
for CF in 3 4 5; do timeout 600 python bench.py --config $CF --no-host-rate --steps 60 > $O/bench_cfg$CF.json 2> $O/bench_cfg$CF.err; done
timeout 300 python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 --no-cpu-baseline --no-pmc --no-host-rate --no-ipopt-leg --no-other-configs --no-distinct > $O/bench_2rank_gloo_selflaunch.json 2> $O/bench_2rank_gloo_selflaunch.err; cut -c1-160 $O/bench_2rank_gloo_selflaunch.json
timeout 600 python tools/options_census.py 2 3 5 > $O/options_census.txt 2>&1; cat $O/options_census.txt
OBCA_HIP_LIBRARY=$C/variants/libobca_hip_hwid.so timeout 300 python tools/determinism_hw.py 12 > $O/determinism_hw.txt 2>&1; grep -E "Uuid: +GPU|differing" $O/determinism_hw.txt
for B in 64 1024; do OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B > $O/phase_B$B.txt 2>&1; OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/phase_profile.py $B ipopt > $O/phase_B${B}_reference_options.txt 2>&1; OBCA_HIP_LIBRARY=$C/libobca_hip_prof.so timeout 200 python tools/quad_gpu.py $B > $O/quad_phase_B$B.txt 2>&1; done
python - <<'PY'
import csv, glob, json
O="gpurun_out/r5final"
for c in ("bench","bench_driver_line","bench_cfg3","bench_cfg4","bench_cfg5","bench_2rank_gloo_selflaunch"):
    try:
        d=json.loads(open(f"{O}/{c}.json").read().strip().splitlines()[-1]); k=d["config"]; r=d.get("roofline") or {}
        print(c,"value",d["value"],"ms",d["ms_per_step"],"kernel_ms",r.get("kernel_ms"),"validated",k.get("converged"),"/",k.get("instances"),"passes",k.get("mean_passes"),"bound",r.get("bound"),r.get("frac"),"impl_hbm",r.get("implementation_hbm_frac"),"io_only",r.get("hbm_io_only_frac"),"traffic",r.get("traffic"),r.get("traffic_over_io_only"),r.get("traffic_over_implementation_model"),"sync",k.get("single_batch_sync_solves_per_s"))
        if k.get("fast_options"): print("   fast", {x: k["fast_options"][x] for x in ("solves_per_s","mean_iterations","mean_passes","solution_differs_from_timed_options","iterations_differ_from_timed_options")})
        if k.get("other_configs"): print("   other", [(o["config"], o["solves_per_s"], o["validated"], o["fast_options"]["solves_per_s"], (o.get("cpu_baseline") or {}).get("value")) for o in k["other_configs"]])
        if k.get("host_pointer"): print("   host", k["host_pointer"]["c_call_solves_per_s"], "distinct", (k.get("distinct_batches") or {}).get("solves_per_s"), "planning", k.get("planning"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(c, "ERR", e)
for d in ("pmc_cfg2","pmc_cfg4","pmc_quad_mfma"):
    v={}
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True) + glob.glob(f"{O}/{d}/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:40]
            if "ipm" in k: v.setdefault(k,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,c in v.items(): print(d, k, {C_:(len(x), sum(x)/len(x)) for C_,x in c.items()})
for f in glob.glob(f"{O}/stats_*/*kernel_stats.csv") + glob.glob(f"{O}/stats_*/**/*kernel_stats.csv", recursive=True): print(f); print(open(f).read()[:600])
PY
