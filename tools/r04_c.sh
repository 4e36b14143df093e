#!/bin/bash
# round 4, run C (one box): accuracy of the rounding-faithful policy's 1 + tanh on the device, its kernel-only A/B, then run B
mkdir -p gpurun_out/r04
timeout -k 10 200 tools/ubench/acc_rf > gpurun_out/r04/acc_rf.txt 2>&1 || echo "acc_rf rc $?"
for r in 1 2 3; do for b in mt_ab_r4tanh_old mt_ab_r4tanh_new; do timeout -k 5 60 tools/ubench/$b 32 30; done; done > gpurun_out/r04/tanh_ab.txt 2>&1
cat gpurun_out/r04/acc_rf.txt gpurun_out/r04/tanh_ab.txt
exec tools/r04_b.sh
