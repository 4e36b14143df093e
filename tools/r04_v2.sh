#!/bin/bash
# round 4, run V2: distribution of 20-tick region times with the two ways of noticing the end (30 regions each, alternating)
mkdir -p gpurun_out/r04
run() { python3 bench.py --steps 20 --warmup 5 --repeats 15 --no-cpu --no-exact-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], ' '.join('%.4f' % x for x in d['wall_ms_per_region']))" "$1"; }
for round in 1 2; do
  FIBHIP_STREAM_WRITE=0 run query
  run write
done > gpurun_out/r04/v2_notice.txt 2>&1
cat gpurun_out/r04/v2_notice.txt
