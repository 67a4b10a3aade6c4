#!/bin/bash
# round 2, job b: the reworked bench.py (default line, 2-rank functional run on one GPU, configs 3/4/5 short)
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
O=gpurun_out/r2b
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -5 $O/bench.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 16 --warmup 4 --batch 512 > $O/bench_2rank.json 2> $O/bench_2rank.err; cat $O/bench_2rank.json; tail -5 $O/bench_2rank.err
timeout 900 python bench.py --config 5 --steps 12 --warmup 4 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; cat $O/bench_cfg5.json; tail -5 $O/bench_cfg5.err
timeout 900 python bench.py --config 4 --steps 8 --warmup 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; cat $O/bench_cfg4.json; tail -5 $O/bench_cfg4.err
timeout 1200 python bench.py --config 3 --steps 12 --warmup 4 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cat $O/bench_cfg3.json; tail -5 $O/bench_cfg3.err
