#!/bin/bash
# round 4, final evidence — part 1 (GPU box): rocprofv3 kernel traces + counter passes of the five bench configurations, the
# measurement table, the chosen plans.  Part 2 (after tools/merge_counters.py has put the counters.json files into
# profiles/counters_r04.json): tools/r04_verify.sh — the -m gpu suite, smoke() and the driver's two bench invocations.
set -o pipefail
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for spec in "fenton512:" "fenton512_exact:--exact" "br512:--model br" "court1024:--model court"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 bash tools/prof.sh r04_$tag $args > gpurun_out/r04/prof_$tag.log 2>&1 || echo "profile $tag failed"
  echo "profile $tag done"
done
FIBHIP_VARIANT=5,54,28,-3 timeout -k 10 500 bash tools/prof.sh r04_fenton4096 --size 4096 --setup 40 --warmup 10 > gpurun_out/r04/prof_fenton4096.log 2>&1 || echo "profile fenton4096 failed"
echo "profile fenton4096 done"
# (the raw per-dispatch traces stay on the box: summaries, kernel statistics and counters are what travels)
for d in gpurun_out/prof_r04_*; do
  find $d -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $d/kernel_stats.csv
  find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
done
timeout -k 10 500 bash tools/bench_table.sh gpurun_out/r04/bench_table.txt > /dev/null 2>&1; cat gpurun_out/r04/bench_table.txt
