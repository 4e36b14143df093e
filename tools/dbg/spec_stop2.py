import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FIBHIP_VARIANT'] = '10,44,25,-3'
from fib_tf_amd import _lib
H, W = 45, 70
rng = np.random.default_rng(5)
init = rng.uniform(0, 1, (4, H, W)).astype(np.float32)
phi = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)

def play(mt, phase, brk, first, getv):
    if mt:
        os.environ.pop('FIBHIP_MT', None)
    else:
        os.environ['FIBHIP_MT'] = '0'
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
    if phase:
        st.set_phase(phi)
    st.set_state(-1, init)
    for _ in range(21):
        st.step(1)
    if brk == 'pace':
        st.pace(H // 4, H // 4 + 5, W // 3, W // 3 + 6, 1.0, 0.0)
    else:
        st.sync()
    st.step(first)
    for _ in range(7 - first):
        st.step(1)
    out = [st.get_state(getv).copy()]
    st.step(3)
    out.append(st.get_state(-1))
    stats = st.launch_stats()
    st.close()
    return out, stats

for phase in (False, True):
    for brk in ('sync', 'pace'):
        for first in (1, 2):
            for getv in (-1, 0):
                a, sa = play(True, phase, brk, first, getv)
                b, sb = play(False, phase, brk, first, getv)
                print('phase', phase, brk, 'first', first, 'get', getv, [bool(np.array_equal(x, y)) for x, y in zip(a, b)],
                      'kept', sa['ahead_stopped_in_time'], 'redone', sa['ahead_recomputed'])
