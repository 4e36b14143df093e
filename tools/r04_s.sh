#!/bin/bash
# round 4, run S: the recovery stress on the final build — random call sequences with a launch giving up anywhere, four configurations
mkdir -p gpurun_out/r04
: > gpurun_out/r04/s_stress.txt
for cfg in "fenton:::7" "br::br:11" "fenton:512::13" "br:512:br:17"; do
  IFS=: read name grid model salt <<< "$cfg"
  echo "== $name grid=${grid:-83x120} salt=$salt" >> gpurun_out/r04/s_stress.txt
  FIBTF_STRESS_SEEDS=${SEEDS:-400} FIBTF_STRESS_SALT=$salt FIBTF_STRESS_GRID=$grid FIBTF_STRESS_MODEL=$model timeout -k 10 420 python -m pytest tests/test_gpu_recovery.py -q -x -k "random_call" -p no:cacheprovider 2>&1 | tail -2 >> gpurun_out/r04/s_stress.txt
done
cat gpurun_out/r04/s_stress.txt
