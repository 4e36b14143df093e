#!/bin/bash
# round 4, final evidence after the bit-identical multiply-add contractions of the rounding-faithful Fenton kernel: its rocprofv3 trace +
# counter passes (tools/merge_counters.py afterwards), then the suite / smoke / bench invocations as the driver runs them
set -o pipefail
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout -k 10 400 bash tools/prof.sh r04_fenton512_exact --exact > gpurun_out/r04/prof_fenton512_exact.log 2>&1 || echo "profile fenton512_exact failed"
d=gpurun_out/prof_r04_fenton512_exact; find $d -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $d/kernel_stats.csv; find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "profile done"
exec bash tools/r04_verify.sh
