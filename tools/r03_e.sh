set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_tick or deferred_ticks or two_handles or stock_library or copy_bandwidth or br_specialised or argument_checks" > gpurun_out/r03/e_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03/e_t.log; tail -4 gpurun_out/r03/e_t.log
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
timeout -k 10 600 bash tools/prof.sh r03_fenton512 > gpurun_out/r03/prof_fenton512.log 2>&1 || exit 1
timeout -k 10 600 bash tools/prof.sh r03_br512 --model br > gpurun_out/r03/prof_br512.log 2>&1 || exit 1
grep -c SIGSEGV gpurun_out/prof_r03_br512/*.err
timeout -k 10 600 bash tools/prof.sh r03_court1024 --model court > gpurun_out/r03/prof_court1024.log 2>&1 || exit 1
FIBHIP_VARIANT=5,54,28,-3 timeout -k 10 900 bash tools/prof.sh r03_fenton4096 --size 4096 --setup 40 --warmup 10 > gpurun_out/r03/prof_fenton4096.log 2>&1 || exit 1
for t in fenton512 br512 court1024 fenton4096; do head -3 gpurun_out/prof_r03_$t/summary.txt | cut -c1-200; done
