#!/bin/bash
# round 4, final evidence after the instruction-count work on the fast policy (Fenton 52 -> 50, Beeler-Reuter 222 -> 202 per cell): the
# rocprofv3 kernel traces + counter passes of the three configurations whose kernels changed (the rounding-faithful Fenton kernel and
# Courtemanche did not), the measurement table.  Then tools/merge_counters.py here, tools/r04_verify.sh on a box.
set -o pipefail
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for spec in "fenton512:" "br512:--model br"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 bash tools/prof.sh r04_$tag $args > gpurun_out/r04/prof_$tag.log 2>&1 || echo "profile $tag failed"
  echo "profile $tag done"
done
FIBHIP_VARIANT=5,54,28,-3 timeout -k 10 500 bash tools/prof.sh r04_fenton4096 --size 4096 --setup 40 --warmup 10 > gpurun_out/r04/prof_fenton4096.log 2>&1 || echo "profile fenton4096 failed"
echo "profile fenton4096 done"
for d in gpurun_out/prof_r04_*; do
  find $d -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $d/kernel_stats.csv
  find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
done
timeout -k 10 500 bash tools/bench_table.sh gpurun_out/r04/bench_table.txt > /dev/null 2>&1; cat gpurun_out/r04/bench_table.txt
