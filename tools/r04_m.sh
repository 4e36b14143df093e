#!/bin/bash
# round 4, run M: Beeler-Reuter's fast step with fewer instructions per cell (FIB_BR_FEWER 1: multiply-adds + bare v_log in the currents,
# 2: + constants folded, 3: + Horner's rule for the twelve sums): pure arithmetic, the kernel, and the single-step error against golden
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in issue_br_bf0 issue_br_bf1 issue_br_bf2 issue_br_bf3; do timeout -k 5 60 ./$b 20000 252 || echo "$b FAILED rc $?"; done
  for b in br_mt_ab_bf0 br_mt_ab_bf1 br_mt_ab_bf2 br_mt_ab_bf3; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done
done > ../../gpurun_out/r04/m_br_fewer.txt 2>&1
cd ../..
for v in 0 1 2 3; do echo "== FIB_BR_FEWER=$v"; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 120 python tools/br_step_error.py 2>&1 | grep 'cheby spec\|worst' | sed -n 1,18p; done > gpurun_out/r04/m_br_fewer_err.txt 2>&1
grep issue_ gpurun_out/r04/m_br_fewer.txt | sort -s -k1,1 | awk '{print $1, $11}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
grep br_mt_ab gpurun_out/r04/m_br_fewer.txt | sort -s -k1,1 | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
cat gpurun_out/r04/m_br_fewer_err.txt
