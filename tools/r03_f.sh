mkdir -p gpurun_out/r03
for s in 400 4000 20000 400 4000 20000; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --setup $s --no-config-legs --no-cpu --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('setup $s', d['value'], d['value_with_snapshots'], d['wall_ms_per_region'], d['launches_per_region'], d['roofline']['us_per_tick'])"
done
