#!/usr/bin/env python3
"""GPU box: what does ONE rank of the weak-scaling bench cost per tick WITHOUT the exchange?  Builds the middle
rank's handle (512 owned rows + ghost rows on both sides) for several halo_ticks and times step() — the compute
price of the communication-avoiding ghost zone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd import _lib
import numpy as np
W, rows, spt = 512, 512, 10
for m in (1, 2, 3, 4, 5, 6, 8):
    g = m * spt
    H = rows + 2 * g
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.5, flags=_lib.FAST | _lib.ROW_INTERLEAVED, global_height=4 * rows,
                      row_offset=rows - g, ghost_top=g, ghost_bottom=g)
    st.set_state(-1, np.random.default_rng(0).uniform(0, 1, (4, H, W)).astype(np.float32))
    st.set_phase(np.random.default_rng(1).uniform(0.5, 1, (H, W)).astype(np.float32))
    st.step(4 * m); st.sync()
    n = 40 * m
    t0 = time.perf_counter(); st.step(n); st.sync(); dt = (time.perf_counter() - t0) / n
    print('halo_ticks %d: ghost %3d rows, slab %4d rows: %.2f us per tick (no exchange), plan %s' % (m, g, H, dt * 1e6, st.launch_plan()))
    st.close()
