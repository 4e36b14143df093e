#!/bin/bash
# round 4, experiment A (one box): kernel-only A/B of the tick boundary without its first barrier (FIB_B_LASTWAVE) for Fenton
# fast / exact and Beeler-Reuter, and of Beeler-Reuter's run-time `skip` made a compile-time constant
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_r4base mt_ab_r4last mt_ab_r4base_exact mt_ab_r4last_exact br_mt_ab_base br_mt_ab_last br_mt_ab_noskip br_mt_ab_noskip_last br_mt_ab_noskip_whole; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done
