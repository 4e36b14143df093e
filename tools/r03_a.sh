set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_tick or deferred_ticks or two_handles or stock_library or copy_bandwidth or br_specialised" > gpurun_out/r03/a_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03/a_t.log; tail -4 gpurun_out/r03/a_t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 bash tools/prof_stock_after_spec.sh || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/prof.sh r03_fenton512 > gpurun_out/r03/prof_fenton512.log 2>&1 || exit 1
timeout -k 10 600 bash tools/prof.sh r03_br512 --model br > gpurun_out/r03/prof_br512.log 2>&1 || exit 1
tail -5 gpurun_out/prof_r03_br512/summary.txt
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_s20.json 2> gpurun_out/r03/bench_s20.err; echo "bench rc=$?"
