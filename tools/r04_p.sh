#!/bin/bash
# round 4, run P: error of Beeler-Reuter's fast policy along the golden trajectories, per FIB_BR_FEWER level (experimental builds of the
# specialised library under tools/ubench/libs), then run O
mkdir -p gpurun_out/r04
for v in 0 1 2 3 4; do echo "== FIB_BR_FEWER=$v"; FIBHIP_BR_LIBRARY=$PWD/tools/ubench/libs/libfibhip_brfewer$v.so timeout -k 5 200 python tools/dbg/br_traj_err.py 2>&1 | tail -12; done > gpurun_out/r04/p_br_traj_err.txt 2>&1
cat gpurun_out/r04/p_br_traj_err.txt
bash tools/r04_o.sh
