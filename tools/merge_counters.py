#!/usr/bin/env python3
"""tools/merge_counters.py OUT.json DIR...: the counters.json records tools/prof_summary.py wrote into each profile directory,
merged into the one file bench.py reads its `roofline.traffic / issue / cache` figures from (profiles/counters_rNN.json)"""
import json
import os
import sys

out, merged = sys.argv[1], {}
for d in sys.argv[2:]:
    path = os.path.join(d, 'counters.json')
    try:
        rec = json.load(open(path))
    except (OSError, ValueError) as e:
        print('skipped %s: %s' % (path, e))
        continue
    for k, v in rec.items():
        v['recorded_in'] = os.path.basename(os.path.normpath(d))
        merged[k] = v
json.dump(merged, open(out, 'w'), indent=1, sort_keys=True)
print('%s: %s' % (out, ', '.join(sorted(merged))))
