#!/bin/bash
# round 4, final evidence, last pass: the Courtemanche profile again (its strip kernel is back on the row-major image), the table's
# Beeler-Reuter / Courtemanche rows, then tools/r04_verify.sh
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout -k 10 400 bash tools/prof.sh r04_court1024 --model court > gpurun_out/r04/prof_court1024.log 2>&1 || echo "profile court1024 failed"
d=gpurun_out/prof_r04_court1024; find $d -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} $d/kernel_stats.csv; find $d -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
run() { python3 bench.py --no-cpu --no-exact-leg --no-config-legs "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-50s %9.0f Mcell-steps/s  %9.3f us/tick  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'] * 1e3, r['frac']))" "$@"; }
{ run --model br --size 512 --steps 3000; run --model br --size 1024 --steps 1000; run --model br --size 2048 --steps 300; run --model court --size 1024 --steps 5000; run --model court --size 2048 --steps 1500; } > gpurun_out/r04/bench_table_tail.txt 2>&1
cat gpurun_out/r04/bench_table_tail.txt
exec tools/r04_verify.sh
