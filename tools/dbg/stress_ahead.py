"""stress: two handles in lockstep — multi-tick launches with series launched ahead / stopped / cancelled against one launch per
tick — random series lengths, observations and pauses (so that the host's word arrives early, just in time and too late);
every observation must be bit-identical.   python tools/dbg/stress_ahead.py [seconds] [size]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd import _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = W = S
rng = np.random.default_rng(12345)
init = np.empty((4, H, W), np.float32)
init[0] = rng.uniform(-0.02, 1.0, (H, W)); init[1:] = rng.uniform(0, 1, (3, H, W))
phi = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
os.environ.pop('FIBHIP_MT', None)
a = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
os.environ['FIBHIP_MT'] = '0'
b = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
os.environ.pop('FIBHIP_MT', None)
for st in (a, b):
    st.set_phase(phi); st.set_state(-1, init)
t0 = time.time(); nobs = 0; nseries = 0
favourite = [10, 20, 5, 32, 12]
while time.time() - t0 < budget:
    # mostly repeated lengths (so that predictions are made), sometimes cut short or overrun
    base = favourite[int(rng.integers(0, len(favourite)))]
    for rep in range(int(rng.integers(1, 5))):
        n = base if rng.random() < 0.6 else int(rng.integers(1, 41))
        for _ in range(n):
            a.step(1); b.step(1)
        nseries += 1
        pause = rng.choice([0.0, 0.0, 2e-5, 1e-4, 1e-3])
        if pause:
            time.sleep(pause)
        op = rng.choice(['sync', 'get', 'getall', 'probe', 'pace', 'set'])
        if op == 'sync':
            a.sync(); b.sync()
        elif op == 'get':
            v = int(rng.integers(0, 4))
            x, y = a.get_state(v).copy(), b.get_state(v).copy()
            assert np.array_equal(x, y), ('get', nseries, float(np.abs(x - y).max()))
            nobs += 1
        elif op == 'getall':
            x, y = a.get_state(-1), b.get_state(-1)
            assert np.array_equal(x, y), ('getall', nseries, float(np.abs(x - y).max()))
            assert np.isfinite(x).all()
            nobs += 1
        elif op == 'probe':
            r, c = int(rng.integers(0, H)), int(rng.integers(0, W))
            assert a.probe(0, r, c) == b.probe(0, r, c), ('probe', nseries)
            nobs += 1
        elif op == 'pace':
            r0, c0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
            a.pace(r0, r0 + 8, c0, c0 + 8, 1.0, 0.0); b.pace(r0, r0 + 8, c0, c0 + 8, 1.0, 0.0)
        else:
            v = int(rng.integers(1, 4))
            z = (b.get_state(v) * np.float32(0.999)).astype(np.float32)
            a.set_state(v, z); b.set_state(v, z)
x, y = a.get_state(-1), b.get_state(-1)
assert np.array_equal(x, y)
print('stress ok: %d series, %d observations compared, %.0f s; %s' % (nseries, nobs, time.time() - t0, a.launch_stats()))
