#!/usr/bin/env python3
"""condense rocprofv3 csv output (kernel trace + pmc passes) into a short per-kernel summary"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split('(')[0]
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
for sub in ('pmc_sq', 'pmc_lds', 'pmc_fetch', 'pmc_write'):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            acc[n][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(n, r['Counter_Name'])] += 1
    if acc:
        print('== %s (per-dispatch averages)' % sub)
    for n, d in acc.items():
        if 'tick_kernel' not in n and 'pointwise' not in n and 'strip_kernel' not in n:
            continue
        print('  ' + n)
        print('     ' + '  '.join('%s=%.4g' % (k, v / cnt[(n, k)]) for k, v in sorted(d.items())))

# machine-readable record for bench.py's roofline.traffic (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
# FETCH_SIZE under-reports wide coalesced reads by 2x per MI355X_MICROARCH.md "HBM", so both the raw and
# the corrected figure are kept)
import json
rec = {}
for sub, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == key:
                acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    for n, v in acc.items():
        if 'strip_kernel' in n or 'tick_kernel' in n:
            rec.setdefault(n, {})[key + '_KiB_per_launch'] = sum(v) / len(v)
for n, d in rec.items():
    f, w = d.get('FETCH_SIZE_KiB_per_launch'), d.get('WRITE_SIZE_KiB_per_launch')
    if f is not None and w is not None:
        d['hbm_bytes_per_launch_raw'] = (f + w) * 1024
        d['hbm_bytes_per_launch_corrected'] = (2 * f + w) * 1024
json.dump(rec, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
