#!/usr/bin/env python3
"""tools/dbg/recovery_seed.py SEED [NTH]: the random call sequence of tests/test_gpu_recovery.py::test_give_up_anywhere_in_random_
call_sequences for one seed, with the op log, under FIBHIP_MT=0 / untouched / FIBHIP_MT_FAKE_GIVEUP=nth: where do they part?"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd import _lib

seed = int(sys.argv[1])
H, W = 83, 120
rng0 = np.random.default_rng(11 * H + W)          # (tests: _state(H, W, 200 + seed))
def _state(H, W, seed, nvar=4):
    rng = np.random.default_rng(seed)
    init = np.empty((nvar, H, W), np.float32)
    init[0] = rng.uniform(-0.02, 1.0, (H, W))
    for v in range(1, nvar):
        init[v] = rng.uniform(0, 1, (H, W))
    return init, rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
init, phi = _state(H, W, 200 + seed)


def play(env):
    for k in ('FIBHIP_MT', 'FIBHIP_MT_FAKE_GIVEUP', 'FIBHIP_AHEAD'):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ['FIBHIP_VARIANT'] = '10,44,25,-3'
    rng = np.random.default_rng(seed)
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
    st.set_phase(phi)
    st.set_state(-1, init)
    seen, log = [], []
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        for _ in range(120):
            op = rng.choice(['step1', 'step1', 'step1', 'step1', 'stepn', 'series', 'pace', 'probe', 'get1', 'getall', 'set1', 'sync',
                             'expect', 'phase'])
            n0 = len(seen)
            arg = ''
            if op == 'step1':
                st.step(1)
            elif op == 'stepn':
                n = int(rng.integers(0, 70)); arg = n
                st.step(n)
            elif op == 'series':
                n = int(rng.integers(2, 12)); arg = n
                for _ in range(3):
                    for _ in range(n):
                        st.step(1)
                    seen.append(st.get_state(0).copy())
            elif op == 'pace':
                r0, c0 = int(rng.integers(0, H - 4)), int(rng.integers(0, W - 4))
                st.pace(r0, r0 + 4, c0, c0 + 4, 1.0, 0.0)
            elif op == 'probe':
                seen.append(np.float32(st.probe(int(rng.integers(0, 4)), int(rng.integers(0, H)), int(rng.integers(0, W)))))
            elif op == 'get1':
                seen.append(st.get_state(int(rng.integers(0, 4))).copy())
            elif op == 'getall':
                seen.append(st.get_state(-1))
            elif op == 'set1':
                v = int(rng.integers(1, 4))
                st.set_state(v, (st.get_state(v) * np.float32(0.999)).astype(np.float32))
            elif op == 'sync':
                st.sync()
            elif op == 'expect':
                n = int(rng.integers(0, 40)); arg = n
                st.expect(n)
            else:
                st.set_phase(phi if rng.integers(0, 2) else None)
            s = st.launch_stats()
            log.append('%-7s %-3s obs %d..%d  ticks %d mt_launches %d fallbacks %s' % (op, arg, n0, len(seen), s['ticks'], s['mt_launches'], st.fallbacks()))
        seen.append(st.get_state(-1))
    stats = st.launch_stats()
    st.close()
    return seen, log, stats


want, _, _ = play({'FIBHIP_MT': '0'})
free, logf, sf = play({})
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 1 + (seed * 7) % max(1, int(sf['mt_launches']))
got, logg, sg = play({'FIBHIP_MT_FAKE_GIVEUP': str(nth)})
def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if not np.array_equal(x, y):
            return i
    return None
print('seed %d: untouched multi-tick run vs one launch per tick: first differing observation %s' % (seed, first_diff(free, want)))
d = first_diff(got, want)
print('seed %d: launch %d of %d gives up: first differing observation %s' % (seed, nth, sf['mt_launches'], d))
if d is not None:
    for line in logg:
        lo, hi = [int(x) for x in line.split('obs ')[1].split()[0].split('..')]
        mark = ' <=== first difference' if lo <= d < hi or (lo == hi == d) else ''
        print('   ' + line + mark)
        if hi > d + 1:
            break
