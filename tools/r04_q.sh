#!/bin/bash
# round 4, run Q: rounding-faithful Fenton kernel with (U - u_0) G as U - U H in one exact multiply-add (FIB_FENTON_FEWER bit 4)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3 4; do for b in mt_ab_ex3 mt_ab_ex19b mt_ab_ex19c; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done; done > ../../gpurun_out/r04/q_exact3.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/q_exact3.txt | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
