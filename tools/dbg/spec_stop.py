import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['FIBHIP_VARIANT'] = '10,44,25,-3'
from fib_tf_amd import _lib
H, W = 45, 70
rng = np.random.default_rng(5)
init = rng.uniform(0, 1, (4, H, W)).astype(np.float32)

def play(mt, cut, L=12, sleep=0.0):
    if mt:
        os.environ.pop('FIBHIP_MT', None)
    else:
        os.environ['FIBHIP_MT'] = '0'
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
    st.set_state(-1, init)
    for _ in range(2):
        for _ in range(L):
            st.step(1)
        st.sync()
    for _ in range(cut):
        st.step(1)
    if sleep:
        time.sleep(sleep)
    out = st.get_state(-1)
    stats = st.launch_stats()
    st.close()
    return out, stats

for cut in (1, 2, 5, 7, 11, 12):
    for sleep in (0.0, 0.02):
        a, sa = play(True, cut, sleep=sleep)
        b, sb = play(False, cut)
        # which tick count does `a` correspond to?
        print('cut', cut, 'sleep', sleep, 'equal', np.array_equal(a, b), 'max|d| %.3g' % np.abs(a - b).max(), sa)
