#!/bin/bash
# one GPU-box job: parity tests, smoke, bench line, rocprofv3 kernel trace + HBM counters.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof -o pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc1.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof -o pmc_write -- python $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc2.err
cd $R; tail -2 gpurun_out/prof.err gpurun_out/pmc1.err gpurun_out/pmc2.err; ls -la gpurun_out/prof; for f in gpurun_out/prof/*kernel_stats*.csv; do head -8 $f; done
for f in gpurun_out/prof/pmc_*counter_collection.csv; do head -3 $f; grep -c . $f; done
