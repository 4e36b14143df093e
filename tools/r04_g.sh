#!/bin/bash
# round 4, run G (one box): round 3's kernels against the working tree's (kernel-only), the stamped build, then the -m gpu suite
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_r03 mt_ab_r4new mt_ab_r03_exact mt_ab_r4new_exact br_mt_ab_base br_mt_ab_new; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/g_ab.txt 2>&1
timeout -k 5 100 ./stamp_mt 8 > ../../gpurun_out/r04/stamp_mt.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/g_ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r04/gputest6.txt 2>&1; tail -4 gpurun_out/r04/gputest6.txt
