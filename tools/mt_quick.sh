set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_tick or deferred_ticks or two_handles or fusion_depths or fenton_trajectory_64 or unit_ops or fenton_single_step or fenton_driver or fenton_ragged" > gpurun_out/r03/mt_t1.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r03/mt_t1.log
tail -5 gpurun_out/r03/mt_t1.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r03/mt_s20.json 2> gpurun_out/r03/mt_s20.err && \
FIBHIP_MT=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-exact-leg > gpurun_out/r03/mt0_s20.json 2> gpurun_out/r03/mt0_s20.err && \
python bench.py --no-cpu --no-exact-leg > gpurun_out/r03/mt_full.json 2> gpurun_out/r03/mt_full.err && \
FIBHIP_MT=0 python bench.py --no-cpu --no-exact-leg > gpurun_out/r03/mt0_full.json 2> gpurun_out/r03/mt0_full.err
python - <<'PY'
import json
for f in ('mt_s20', 'mt0_s20', 'mt_full', 'mt0_full'):
    try:
        d = json.load(open('gpurun_out/r03/%s.json' % f))
        r = d['roofline']
        print(f, 'value', d['value'], 'ms/tick', d['ms_per_step'], 'us/launch', r['us_per_launch'], 'launches', r['launches_timed'], 'ticks', r['ticks_timed'], 'frac', r['frac'], 'snap', d.get('value_with_snapshots'), 'walls', d['wall_ms_per_region'])
    except Exception as e:
        print(f, 'failed', e)
PY
