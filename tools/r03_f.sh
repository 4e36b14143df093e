set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_ahead or multi_tick or deferred_ticks or two_handles or readback or driver_semantics or screen or timeline" > gpurun_out/r03/f_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03/f_t.log; tail -12 gpurun_out/r03/f_t.log
[ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu --no-config-legs --no-exact-leg > gpurun_out/r03/f_default.json 2> gpurun_out/r03/f_default.err; echo rc=$?
python bench.py --steps 20 --warmup 5 --no-cpu --no-config-legs --no-exact-leg > gpurun_out/r03/f_s20.json 2> gpurun_out/r03/f_s20.err; echo rc=$?
FIBHIP_AHEAD=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-config-legs --no-exact-leg > gpurun_out/r03/f_s20_noahead.json 2> /dev/null; echo rc=$?
python - <<'PY'
import json
for f in ('f_default', 'f_s20', 'f_s20_noahead'):
    d = json.load(open('gpurun_out/r03/%s.json' % f))
    print(f, 'value', d['value'], 'snap', d['value_with_snapshots'], 'ratio %.3f' % (d['value_with_snapshots'] / d['value']), 'walls', d['wall_ms_per_region'])
PY
