mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_ahead or read_back or multi_tick or deferred or two_handles" > gpurun_out/r03/j_t.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03/j_t.log
python bench.py --no-cpu --no-config-legs --no-exact-leg > gpurun_out/r03/j_default.json 2> /dev/null
python tools/dbg/gc_pause.py nogc 2>&1 | tail -2
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03/j_default.json'))
print('value', d['value'], 'snap', d['value_with_snapshots'], 'ratio %.3f' % (d['value_with_snapshots'] / d['value']), 'us/tick', d['roofline']['us_per_tick'])
PY
