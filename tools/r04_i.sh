#!/bin/bash
# round 4, run I: the window's centre column from the lane's own registers where no clamp applies (FIB_OWN_CENTRE)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3 4; do
  for b in mt_ab_r03 mt_ab_cur mt_ab_oc mt_ab_cur_exact mt_ab_oc_exact; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/i_own_centre.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/i_own_centre.txt | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
