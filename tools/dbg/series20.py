import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd import _lib
H = W = 512
rng = np.random.default_rng(5)
init = rng.uniform(0, 1, (4, H, W)).astype(np.float32)
phi = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST)
st.set_phase(phi)
st.set_state(-1, init)
st.step(1); st.sync()
for _ in range(300):
    st.step(1)
import gc; gc.collect(); gc.freeze()
for n in (20, 20, 20, 20, 10, 10, 10, 32, 32, 32, 5, 5, 5):
    st.sync()
    s0 = st.launch_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        st.step(1)
    t1 = time.perf_counter()
    st.sync()
    t2 = time.perf_counter()
    s1 = st.launch_stats()
    print('series %2d: %.1f us (%.2f per tick), host loop %.1f us, launches %d, stats %s' % (
        n, (t2 - t0) * 1e6, (t2 - t0) * 1e6 / n, (t1 - t0) * 1e6, s1['launches'] - s0['launches'], s1))
