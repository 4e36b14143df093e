#!/bin/bash
# round 4, run U: Beeler-Reuter with each four of eight reciprocals from ONE v_rcp_f32 (FIB_BR_FEWER=3) against the shipped level 2:
# pure arithmetic, the kernel alone, the distance from the golden trajectories, the full-size oracle test
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in issue_br_bf2 issue_br_bf3; do timeout -k 5 60 ./$b 20000 252 || echo "$b FAILED rc $?"; done
  for b in br_mt_ab_bf2 br_mt_ab_bf3; do timeout -k 5 60 ./$b 32 40 || echo "$b FAILED rc $?"; done
done > ../../gpurun_out/r04/u_br_rcp.txt 2>&1
cd ../..
for v in 2 3; do echo "== FIB_BR_FEWER=$v"; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 200 python tools/dbg/br_traj_err.py 2>&1 | grep -v elapsed | tail -6; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 300 python -m pytest tests/test_gpu_fullsize.py -q -s -k "br and fast" -p no:cacheprovider 2>&1 | grep 'br 512\|passed\|failed'; done > gpurun_out/r04/u_br_rcp_err.txt 2>&1
grep issue_ gpurun_out/r04/u_br_rcp.txt | sed 's/:.*steps: /: /' | cut -c1-60
grep br_mt_ab gpurun_out/r04/u_br_rcp.txt | sort -s -k1,1 | awk '{print $1, $(NF-2)}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}' | sort
cat gpurun_out/r04/u_br_rcp_err.txt
