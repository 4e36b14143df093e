#!/bin/bash
# round 4, run K: the instruction-issue bound of one sub-step, measured (tools/ubench/issue_bound.hip): the kernels' own device
# functions on registers, no LDS / barrier / memory, 4 ... 16 waves per workgroup, one workgroup per compute unit
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2; do
  for b in issue_fenton_w4 issue_fenton_w8 issue_fenton_w12 issue_fenton_w15 issue_fenton_w16 issue_fenton_exact issue_br_w4 issue_br_w8 issue_br_w12 issue_br_w15 issue_br_w16 issue_br_exact; do
    timeout -k 5 60 ./$b 20000 252 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/k_issue_bound.txt 2>&1
cd ../..
cat gpurun_out/r04/k_issue_bound.txt
