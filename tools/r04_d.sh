#!/bin/bash
# round 4, run D: the pushed-epoch ("inbox") boundary against the polled one, kernels alone; the two-library reproducer
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_r4base mt_ab_r4inbox br_mt_ab_base br_mt_ab_inbox; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/d_inbox.txt 2>&1
cd ../..
cat gpurun_out/r04/d_inbox.txt
tools/repro_two_libs/run.sh > gpurun_out/r04/repro_two_libs.txt 2>&1; grep -v simple_timer gpurun_out/r04/repro_two_libs.txt | tail -30
