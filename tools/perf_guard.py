#!/usr/bin/env python3
"""GPU box: run the bench configurations whose results are committed under profiles/ and flag any that lost more
than 4 % — run after every change to csrc/ (a uniform run-time branch in tick_kernel once cost 8-10 % unnoticed).

    python tools/perf_guard.py            # compares with profiles/r01_perf_guard.json
    python tools/perf_guard.py --update   # rewrites it"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'profiles', 'r01_perf_guard.json')
CASES = {
    'fenton512': [], 'fenton512_exact': ['--exact'], 'fenton1024': ['--size', '1024'], 'fenton2048': ['--size', '2048'],
    'br512': ['--model', 'br'], 'br2048': ['--model', 'br', '--size', '2048', '--steps', '300'],
    'court1024': ['--model', 'court', '--size', '1024'], 'court512': ['--model', 'court', '--size', '512'],
}


def run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu'] + args, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)['value']


def main():
    got = {k: run(a) for k, a in CASES.items()}
    if '--update' in sys.argv or not os.path.exists(REF):
        json.dump(got, open(REF, 'w'), indent=1)
        print('wrote', REF)
    ref = json.load(open(REF))
    bad = 0
    for k, v in got.items():
        r = ref.get(k)
        flag = '' if r is None or v >= 0.96 * r else '   <-- REGRESSION'
        bad += bool(flag)
        print('%-16s %10.0f Mcell-steps/s   (reference %s)%s' % (k, v, '%.0f' % r if r else 'none', flag))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
