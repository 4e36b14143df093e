#!/usr/bin/env python3
"""Two PROCESSES, each stepping a 512x512 multi-tick grid (252 tiles that want every compute unit) on the same GPU, wait bound 1 ms:
when both launches start together the dispatcher interleaves their workgroups, neither grid is fully resident, tiles wait out their
bound and give up FOR REAL — the handles recover and must end with the bits of the one-launch-per-tick run.
    python tools/dbg/two_processes_giveup.py            (parent: starts the partner, compares)"""
import os
import subprocess
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def state(seed):
    rng = np.random.default_rng(seed)
    init = np.empty((4, 512, 512), np.float32)
    init[0] = rng.uniform(-0.02, 1.0, (512, 512))
    for v in range(1, 4):
        init[v] = rng.uniform(0, 1, (512, 512))
    return init, rng.uniform(0.3, 1.0, (512, 512)).astype(np.float32)


def run(mt, seconds, seed):
    from fib_tf_amd import _lib
    os.environ.pop('FIBHIP_MT', None)
    if mt:
        os.environ['FIBHIP_MT_WAIT_MS'] = '1'
    else:
        os.environ['FIBHIP_MT'] = '0'
    init, phi = state(seed)
    st = _lib.Stepper(_lib.FENTON4V, 512, 512, 0.1, 1.3, flags=_lib.FAST)
    st.set_phase(phi)
    st.set_state(-1, init)
    st.step(1)
    st.sync()
    t_end = time.time() + seconds
    n = 0
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        while time.time() < t_end if seconds else n < run.ticks:
            st.step(32)
            n += 32
            if n % 256 == 0:
                st.sync()
        out = st.get_state(-1)
    fb = st.fallbacks()
    st.close()
    return out, fb, n


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'partner':
        out, fb, n = run(True, float(sys.argv[2]), 5)
        run.ticks = n
        want, _, _ = run(False, 0, 5)
        same = bool(np.array_equal(out, want))
        print('partner: %d ticks, gave up %d (recomputed %d ticks); equal to one launch per tick: %s' % (n, fb[0], fb[1], same), flush=True)
        sys.exit(0 if same else 1)
    partner = subprocess.Popen([sys.executable, os.path.abspath(__file__), 'partner', '6'])
    time.sleep(2.0)                                   # (the partner's import + first launches)
    got, fb, n = run(True, 2.0, 9)
    prc = partner.wait(timeout=120)
    run.ticks = n
    want, _, _ = run(False, 0, 9)
    print('this process: %d ticks beside the partner, gave up %d (recomputed %d ticks); equal to one launch per tick: %s'
          % (n, fb[0], fb[1], np.array_equal(got, want)))
    sys.exit(0 if np.array_equal(got, want) and prc == 0 else 1)
