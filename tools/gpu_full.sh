# the whole -m gpu suite, then one bench line per model (round-3 working script)
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03/full_t.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r03/full_t.log
tail -6 gpurun_out/r03/full_t.log
[ $rc -eq 0 ] || exit $rc
for m in fenton br court; do
  FIBHIP_PRINT_PLAN=1 python bench.py --model $m --no-cpu --no-exact-leg > gpurun_out/r03/b_$m.json 2> gpurun_out/r03/b_$m.err || exit 1
done
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r03/b_s20.json 2> gpurun_out/r03/b_s20.err
python - <<'PY'
import json
for f in ('b_fenton', 'b_br', 'b_court', 'b_s20'):
    try:
        d = json.load(open('gpurun_out/r03/%s.json' % f))
        r = d['roofline']
        print(f, 'value', d['value'], 'ms/tick', d['ms_per_step'], 'events us/tick %.3f' % (r['us_per_launch'] * r['launches_timed'] / r['ticks_timed']), 'launches', r['launches_timed'], 'ticks', r['ticks_timed'], 'frac', r['frac'], 'snap', d.get('value_with_snapshots'))
    except Exception as e:
        print(f, 'failed', e)
PY
grep -h fibhip gpurun_out/r03/b_*.err | sort | uniq -c
