#!/bin/bash
# what the driver runs at round end, on one box: the -m gpu suite, smoke(), the two bench invocations
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04/verify_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r04/verify_t.log; tail -3 gpurun_out/r04/verify_t.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/verify_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r04/verify_smoke.log
python bench.py > gpurun_out/r04/final_bench.json 2> gpurun_out/r04/final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/final_bench_s20.json 2> gpurun_out/r04/final_bench_s20.err; echo "bench s20 rc=$?"
