#!/bin/bash
# round 4, run H: semantically equivalent builds of the multi-tick kernels (a: shift-form wait bound + Geo normalised, b: full Geo,
# c: multiply-form bound + Geo normalised, d: multiply-form + full Geo) against round 3's kernel — the kernels sit at their register
# limits and a change anywhere moves them by 2-3 %
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3 4; do
  for b in mt_ab_r03 mt_ab_n1 mt_ab_n5 mt_ab_n6 mt_ab_r03_exact mt_ab_n1_exact mt_ab_n5_exact mt_ab_n6_exact br_mt_ab_base br_mt_ab_n1 br_mt_ab_n5 br_mt_ab_n6; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/h_variants.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/h_variants.txt | awk '{print $1, $(NF-4), $(NF-2)}' | awk '{a[$1]=a[$1]" "$3} END{for(k in a) print k, a[k]}' | sort
