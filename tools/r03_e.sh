# round 3, second session: the multi-tick / run-ahead tests with the whole-series launch and the stop word, then the driver's
# bench invocation twice and the default one
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_tick or deferred or ahead or read_back or two_handles or series or launch_count" > gpurun_out/r03/e_t.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r03/e_t.log; tail -5 gpurun_out/r03/e_t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-cpu > gpurun_out/r03/e_s20.json 2> gpurun_out/r03/e_s20.err; echo "bench s20 rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-cpu > gpurun_out/r03/e_s20b.json 2>> gpurun_out/r03/e_s20.err; echo "bench s20 rc=$?"
FIBHIP_AHEAD=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs --no-cpu > gpurun_out/r03/e_s20_noahead.json 2>> gpurun_out/r03/e_s20.err; echo "bench s20 noahead rc=$?"
timeout -k 10 400 python bench.py --no-config-legs --no-cpu > gpurun_out/r03/e_default.json 2> gpurun_out/r03/e_default.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ('e_s20','e_s20b','e_s20_noahead','e_default'):
    try:
        d=json.loads(open('gpurun_out/r03/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["value_with_snapshots"], d["wall_ms_per_region"], d.get("launches_per_region"), d['roofline']['us_per_tick'], d['config']['launch_stats'])
    except Exception as e:
        print(f, 'unreadable', e)
P
