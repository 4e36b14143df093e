#!/bin/bash
# round 4, run V: the end of a launch noticed through a word the stream writes into host memory (default now) against hipStreamQuery
# (FIBHIP_STREAM_WRITE=0): the driver's own invocation, three alternating rounds, one box; then the recovery / parity tests that lean on sync
mkdir -p gpurun_out/r04
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu --no-exact-leg --no-config-legs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s value %.0f ms_per_step %.5f regions %s snapshots %.0f' % (sys.argv[1], d['value'], d['ms_per_step'], d['wall_ms_per_region'], d.get('value_with_snapshots') or 0))" "$1"; }
for round in 1 2 3; do
  FIBHIP_STREAM_WRITE=0 run query
  run write
done > gpurun_out/r04/v_notice.txt 2>&1
cat gpurun_out/r04/v_notice.txt
timeout -k 10 600 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_parity.py -q -x -k "recover or give or series or run_ahead or read_back or deferred or two_handles or expect or declared" -p no:cacheprovider 2>&1 | tail -3
