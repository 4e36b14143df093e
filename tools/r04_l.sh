#!/bin/bash
# round 4, run L: Fenton's fast reaction term with fewer instructions per cell (FIB_FENTON_FEWER bit 0: U*G(b) as U - U*H(b), bit 1:
# the second sigmoid itself instead of one minus its complement): the multi-tick kernel alone, four rounds
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3 4; do
  for b in mt_ab_fw0 mt_ab_fw3 mt_ab_fw3h; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done
done > ../../gpurun_out/r04/l_fewer3.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/l_fewer3.txt | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
