# round 3, second session — evidence for the final kernels: rocprofv3 kernel trace + counter passes of the three configs whose
# kernel changed (multi-tick Fenton fast / rounding-faithful, Beeler-Reuter), the tick-boundary stamps, the series timings
set -o pipefail
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for spec in "fenton512:" "fenton512_exact:--exact" "br512:--model br"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 600 bash tools/prof.sh r03b_$tag $args > gpurun_out/r03/prof_b_$tag.log 2>&1 || echo "profile $tag failed"
  tail -3 gpurun_out/r03/prof_b_$tag.log
done
timeout 120 ./tools/ubench/stamp_mt 8 > gpurun_out/r03/stamp_mt_b.txt 2>&1; head -3 gpurun_out/r03/stamp_mt_b.txt
timeout -k 10 200 python tools/dbg/series20.py > gpurun_out/r03/series20.txt 2>&1; tail -4 gpurun_out/r03/series20.txt
