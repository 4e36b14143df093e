#!/bin/bash
# the DESIGN.md measurement table: bench.py on one MI355X for every model / size of the table (fast + exact)
out=${1:-gpurun_out/bench_table.txt}
: > $out
run() { python3 bench.py --no-cpu --no-exact-leg --no-config-legs "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('%-64s %9.0f Mcell-steps/s  %9.3f us/tick  K=%-2d x%d%s  tile %-26s frac %.3f  us/launch %.2f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'] * 1e3, d['config']['fused_sub_steps_per_launch'], d['config']['launches_per_tick'], (', %g ticks per launch' % r['ticks_per_launch']) if r.get('ticks_per_launch', 1) > 1 else '', d['config']['tile'].split(' (')[0], r['frac'], r['us_per_launch']))" "$@" >> $out; }
run --model fenton --size 512 --steps 5000
run --model fenton --size 512 --steps 5000 --exact
run --model fenton --size 768 --steps 3000
run --model fenton --size 1024 --steps 2000
run --model fenton --size 2048 --steps 600
run --model fenton --size 4096 --steps 200
run --model fenton --size 4096 --steps 200 --exact
run --model br --size 512 --steps 3000
run --model br --size 512 --steps 3000 --exact
run --model br --size 1024 --steps 1000
run --model br --size 2048 --steps 300
run --model court --size 1024 --steps 5000
run --model court --size 1024 --steps 5000 --exact
run --model court --size 512 --steps 5000
run --model court --size 2048 --steps 1500
cat $out
