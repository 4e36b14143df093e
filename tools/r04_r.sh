#!/bin/bash
# round 4, run R: Courtemanche's fast tick on aggregates with its constants kept in registers and a branch-free m-gate (FIB_COURT_FEWER)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do for b in court_ab_cf0 court_ab_cf1; do timeout -k 5 60 ./$b 40 9 || echo "$b FAILED rc $?"; done; done > ../../gpurun_out/r04/r_court.txt 2>&1
cd ../..
cat gpurun_out/r04/r_court.txt
