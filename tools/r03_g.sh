mkdir -p gpurun_out/r03
for m in "fenton --exact" "br --exact" "court --exact" "br"; do
  tag=$(echo $m | tr -d ' -')
  FIBHIP_PRINT_PLAN=1 python bench.py --model $m --no-cpu --no-exact-leg --no-config-legs --repeats 3 --steps 1000 > gpurun_out/r03/x_$tag.json 2> gpurun_out/r03/x_$tag.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03/x_$tag.json')); r = d['roofline']
print('$m', 'value', d['value'], 'us/tick', r['us_per_tick'], 'frac', r['frac'], r['kernel'], d['config']['tile'])
PY
done
grep -h "fibhip:" gpurun_out/r03/x_*.err | sort | uniq
