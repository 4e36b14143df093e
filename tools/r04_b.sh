#!/bin/bash
# round 4, run B (one box): the driver's own invocation, the sharded bench rehearsals with their side legs, the whole -m gpu suite
mkdir -p gpurun_out/r04
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_steps20.json 2> gpurun_out/r04/bench_steps20.err || echo "bench steps20 rc $?"
FIBTF_ONE_DEVICE=1 FIBTF_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 8 --warmup 4 --setup 8 --no-cpu > gpurun_out/r04/bench_gpus2.json 2> gpurun_out/r04/bench_gpus2.err || echo "bench gpus2 rc $?"
FIBTF_ONE_DEVICE=1 FIBTF_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 8 --warmup 4 --setup 8 --no-cpu > gpurun_out/r04/bench_gpus4.json 2> gpurun_out/r04/bench_gpus4.err || echo "bench gpus4 rc $?"
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/r04/gputest4.txt 2>&1
tail -15 gpurun_out/r04/gputest4.txt
