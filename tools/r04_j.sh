#!/bin/bash
# round 4, run J: the reaction term issued BEFORE the wait for the window (FIB_PRE_FIRST 1: sched_barrier, 2: ordered asm statements,
# 3: + the state opaque behind the barrier so that the term cannot move in front of it) against the shipped order (0)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_pf0 mt_ab_pf1 mt_ab_pf2 mt_ab_pf3 mt_ab_pf0_exact mt_ab_pf1_exact mt_ab_pf2_exact mt_ab_pf3_exact; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/j_prefirst.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/j_prefirst.txt | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
