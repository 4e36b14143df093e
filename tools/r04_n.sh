#!/bin/bash
# (levels 3 and 4 were measured and removed again — 4-10 x further from the golden trajectories, profiles/r04_fewer_instructions.txt — the tree has 0 .. 2)
# round 4, run N: FIB_BR_FEWER=4 (x as one multiply-add, iK1's first quotient with one instruction fewer) against 3 and 0
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in issue_br_bf0 issue_br_bf3 issue_br_bf4; do timeout -k 5 60 ./$b 20000 252 || echo "$b FAILED rc $?"; done
  for b in br_mt_ab_bf0 br_mt_ab_bf3 br_mt_ab_bf4; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done
done > ../../gpurun_out/r04/n_br_fewer.txt 2>&1
cd ../..
for v in 0 3 4; do echo "== FIB_BR_FEWER=$v"; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 120 python tools/br_step_error.py 2>&1 | grep 'cheby spec' | sed -n 1,8p; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 300 python -m pytest tests/test_gpu_fullsize.py -q -s -k "br" -p no:cacheprovider 2>&1 | grep -i 'err\|measured\|passed\|failed' | cut -c1-220; done > gpurun_out/r04/n_br_fewer_err.txt 2>&1
grep issue_ gpurun_out/r04/n_br_fewer.txt | sed 's/:.*steps: /: /' | cut -c1-50
grep br_mt_ab gpurun_out/r04/n_br_fewer.txt | sort -s -k1,1 | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
cat gpurun_out/r04/n_br_fewer_err.txt
