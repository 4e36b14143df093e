#!/bin/bash
# round 4, run F (one box): did this round's kernel edits cost the headline kernel anything (round 3's kernels.hpp against HEAD's
# and the working tree's), and what does the paired LDS image buy three-row strips (two copies of the tile program by strip parity)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_r03 mt_ab_r4head mt_ab_r4p0 mt_ab_r4p1 mt_ab_r4p1m mt_ab_r4p0_exact mt_ab_r4p1_exact br_mt_ab_base br_mt_ab_pair; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/f_pair3.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/f_pair3.txt
