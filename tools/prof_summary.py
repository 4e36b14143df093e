#!/usr/bin/env python3
"""condense rocprofv3 csv output (kernel trace + pmc passes of tools/prof.sh) into a short per-kernel summary
(stdout) and a machine-readable record of the dominant kernel (counters.json) that bench.py reads back for
`roofline.traffic`, `roofline.issue` and `roofline.cache`"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
PASSES = ('pmc_sq', 'pmc_lds', 'pmc_trans', 'pmc_fetch', 'pmc_write', 'pmc_l2', 'pmc_ea', 'pmc_eaw')


def short(name):
    import re
    name = name.split('(')[0]
    name = re.sub(r'b_libfibhip_[0-9A-Za-z_]*::', '', name)         # the build tag of a specialised library (csrc/models.hpp)
    for a, b in (('void fib::', ''), ('fib::', ''), ('(fib::PhaseTab)', '')):
        name = name.replace(a, b)
    return name[:110]


rows = []
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
dur = defaultdict(list)
meta = {}
for r in rows:
    n = short(r['Kernel_Name'])
    dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    meta[n] = (r.get('VGPR_Count', '?'), r.get('SGPR_Count', '?'), r.get('LDS_Block_Size', '?'), r.get('Grid_Size', '?'),
               r.get('Workgroup_Size', '?'))
print('== kernel trace (rocprofv3 --kernel-trace): per-kernel durations')
tot = sum(sum(v) for v in dur.values()) or 1
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('%-112s calls %6d  avg %9.2f us  med %9.2f us  min %8.2f  max %9.2f  %5.1f%%  vgpr/sgpr/lds/grid/wg %s'
          % (n, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3, v2[-1] / 1e3, 100.0 * sum(v) / tot,
             '/'.join(map(str, meta[n]))))
ours = [n for n in dur if 'tick_kernel' in n or 'strip_kernel' in n or 'strip_mt_kernel' in n or 'pointwise' in n]
dominant = max(ours, key=lambda n: sum(dur[n])) if ours else None


def bench_line(name):
    """the bench line printed under one pass (launch_stats: how many ticks the multi-tick launches of the pass advanced)"""
    try:
        line = [l for l in open(os.path.join(out, name)) if l.startswith('{')][-1]
        return json.loads(line)
    except (OSError, IndexError, ValueError):
        return {}


def mt_ticks(name):
    return bench_line(name).get('config', {}).get('launch_stats', {}).get('mt_ticks', 0)


# A multi-tick kernel's dispatches advance 1 ... 32 ticks each: its counters are recorded PER TICK (sum over the pass's
# dispatches / ticks those advanced, from the bench line of the same pass)
per_tick = bool(dominant and 'strip_mt_kernel' in dominant)
avg = defaultdict(dict)            # kernel -> counter -> per-dispatch average (per-tick for the multi-tick kernel)
for sub in PASSES:
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    ticks = mt_ticks(sub + '.json') if per_tick else 0
    for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            acc[n][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(n, r['Counter_Name'])] += 1
    if acc:
        print('== %s (per-dispatch averages)' % sub)
    for n, d in acc.items():
        if n not in ours:
            continue
        print('  ' + n)
        print('     ' + '  '.join('%s=%.6g' % (k, v / cnt[(n, k)]) for k, v in sorted(d.items())))
        for k, v in d.items():
            avg[n][k] = v / cnt[(n, k)]
        if per_tick and n == dominant and ticks > 0:
            print('     per tick (%d ticks in %d dispatches): ' % (ticks, max(cnt[(n, k)] for k in d)) +
                  '  '.join('%s=%.6g' % (k, v / ticks) for k, v in sorted(d.items())))
            for k, v in d.items():
                avg[n][k] = v / ticks

# machine-readable record of the dominant kernel.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
# half of a wide coalesced read stream (MI355X_MICROARCH.md "HBM"), so raw and corrected figures are both kept
rec = {}
if dominant:
    d = dict(avg.get(dominant, {}))
    d['kernel'] = dominant
    d['us_per_launch_under_trace'] = sum(dur[dominant]) / len(dur[dominant]) / 1e3
    import re
    mm = re.search(r'<[^,]+, [^,]+, \d+, (\d+), (\d+), (\d+), (\d+)', dominant)
    if mm:                                           # K, TX, TY, R (strips) or threads (flat tiles)
        r = int(mm.group(4))
        d['tile'] = [int(mm.group(2)), int(mm.group(3)), r if 'strip' in dominant or 'rows_kernel' in dominant else -r]
    if per_tick:
        t = mt_ticks('bench_trace.json')
        d['per'] = 'tick'
        if t > 0:
            d['us_per_tick_under_trace'] = sum(dur[dominant]) / t / 1e3
            d['ticks_per_launch_under_trace'] = t / len(dur[dominant])
    f, w = d.get('FETCH_SIZE'), d.get('WRITE_SIZE')
    if f is not None and w is not None:
        d['FETCH_SIZE_KiB_per_launch'], d['WRITE_SIZE_KiB_per_launch'] = f, w
        d['hbm_bytes_per_launch_raw'] = (f + w) * 1024             # (per tick when d['per'] == 'tick')
        d['hbm_bytes_per_launch_corrected'] = (2 * f + w) * 1024
    key = None
    try:
        line = [l for l in open(os.path.join(out, 'bench_trace.json')) if l.startswith('{')][-1]
        key = json.loads(line)['roofline'].get('counters_key')
    except (OSError, IndexError, ValueError, KeyError):
        pass
    rec[key or dominant] = d
json.dump(rec, open(os.path.join(out, 'counters.json'), 'w'), indent=1)
print('== counters.json key:', list(rec))
