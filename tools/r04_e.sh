#!/bin/bash
# round 4, run E (one box): the paired-row LDS image (ds_read_b64 windows) — Fenton with four-row strips against the shipped
# three-row strips, Beeler-Reuter's two-row strips against their ds_read_b32 form; merged (ds_read2st64_b64) against separate reads
mkdir -p gpurun_out/r04
cd tools/ubench
for round in 1 2 3; do
  for b in mt_ab_r4base mt_ab_r4_R4_25 mt_ab_r4_R4_25_merged mt_ab_r4_R4_26 mt_ab_r4_R4_28 br_mt_ab_base br_mt_ab_pair br_mt_ab_pair_merged; do
    timeout -k 5 60 ./$b 32 30 || echo "$b FAILED rc $?"
  done
done > ../../gpurun_out/r04/e_pair.txt 2>&1
cd ../..
sort -s -k1,1 gpurun_out/r04/e_pair.txt
