# the driver's invocation three times: value, value_with_snapshots, launches and wall time per region of both legs
mkdir -p gpurun_out/r03
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-cpu --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['value_with_snapshots'], d['wall_ms_per_region'], d['launches_per_region'], d['snapshots_regions'], d['config']['launch_stats'])"
done
