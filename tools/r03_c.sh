export TMPDIR=/tmp
out=$PWD/gpurun_out/r03/sigsegv2
mkdir -p $out
for mode in plain big phase image timed pace plan single phase_image_timed_pace_plan_single_big; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode -- python3 tools/dbg/sas_variants.py $mode > $out/$mode.out 2> $out/$mode.err
  echo "$mode rc=$? $(grep -c SIGSEGV $out/$mode.err)"
done
