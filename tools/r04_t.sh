#!/bin/bash
# round 4, run T: does the arithmetic get cheaper per instruction with more waves per SIMD?  One-cell-per-lane variant of the pure loop
# (52 registers), 16 waves per workgroup: one workgroup per compute unit (4 waves per SIMD) against two (8 per SIMD)
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2; do
  timeout -k 5 60 ./issue_occ_r1_w16_1 20000 252
  timeout -k 5 60 ./issue_occ_r1_w16_2 20000 504
  timeout -k 5 60 ./issue_occ_r2_w16_1 20000 252
  timeout -k 5 60 ./issue_occ_r2_w8_4 20000 504
done > ../../gpurun_out/r04/t_occupancy.txt 2>&1
cd ../..
cut -c1-170 gpurun_out/r04/t_occupancy.txt
