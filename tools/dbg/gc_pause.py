"""GPU box: one image() in ~1350 takes 40 ms instead of 0.16 — the interpreter's cyclic garbage collector (mode `nogc`: none;
`noahead` with FIBHIP_AHEAD=0: the same stall without run-ahead).  Prints per-repetition totals and every image() over 1 ms."""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fib_tf_amd.fenton import Fenton4v
m = Fenton4v({'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000, 'diff': 1.5, 'fast_math': True})
m.add_hole_to_phase_field(256, 256, 30)
m.define()
st = m._stepper
st.step(500); st.sync()
mode = sys.argv[1] if len(sys.argv) > 1 else ''
if 'nogc' in mode:
    gc.disable()
for rep in range(6):
    m.image(); st.sync()
    img = []
    t0 = time.perf_counter()
    for tick in range(3000):
        st.step(1)
        if tick % 10 == 0:
            b = time.perf_counter()
            m.image()
            img.append(time.perf_counter() - b)
    st.sync()
    img = np.array(img) * 1e6
    print(mode, 'rep', rep, 'total %.1f ms' % ((time.perf_counter() - t0) * 1e3), 'image mean %.1f' % img.mean(), 'over 1 ms:', [(int(i), round(float(v) / 1e3, 1)) for i, v in enumerate(img) if v > 1000])
