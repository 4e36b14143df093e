# round 3, final evidence — part 1 (GPU box): the whole -m gpu suite, the profiles, the measurement table, the chosen plans.
# Part 2 (after the profiles' counters.json files have been merged into profiles/counters_r03.json): tools/r03_verify.sh runs
# the suite again, smoke() and the two bench invocations of the driver, so that the bench lines quote the final counters.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/final_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03/final_t.log; tail -4 gpurun_out/r03/final_t.log
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
for spec in "fenton512:" "fenton512_exact:--exact" "br512:--model br" "court1024:--model court"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 600 bash tools/prof.sh r03_$tag $args > gpurun_out/r03/prof_$tag.log 2>&1 || echo "profile $tag failed"
done
FIBHIP_VARIANT=5,54,28,-3 timeout -k 10 900 bash tools/prof.sh r03_fenton4096 --size 4096 --setup 40 --warmup 10 > gpurun_out/r03/prof_fenton4096.log 2>&1 || echo "profile fenton4096 failed"
timeout -k 10 900 bash tools/bench_table.sh gpurun_out/r03/bench_table.txt > /dev/null 2>&1; cat gpurun_out/r03/bench_table.txt
timeout -k 10 600 bash tools/chosen_plans.sh > gpurun_out/r03/chosen_plans.txt 2>&1; wc -l gpurun_out/r03/chosen_plans.txt
timeout 120 ./tools/ubench/stamp_mt 8 > gpurun_out/r03/stamp_mt.txt 2>&1
