#!/usr/bin/env python3
"""GPU box: cost of the host-buffer legs of the C ABI (set_state / get_state over PCIe through the pinned staging
buffer) next to the stepping itself, Fenton 512x512: what a job pays if it uploads a state, runs 1 s of simulated
time (1000 ticks) and downloads the result, and what one image() snapshot costs"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.fenton import Fenton4v  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = Fenton4v({'height': N, 'width': N, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 1.5, 'duration': 1000})
m.add_hole_to_phase_field(N // 2, N // 2, N // 17)
m.define()
st = m._stepper
full = st.get_state(-1)
st.step(50); st.sync()


def best(f, n=20):
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return min(t)


up = best(lambda: (st.set_state(-1, full), st.sync()))
down = best(lambda: st.get_state(-1))
img = best(lambda: m.image())
run = best(lambda: (st.step(1000), st.sync()), 3)
cells = N * N * 10000
print('Fenton %dx%d: set_state(all 4 arrays) %.0f us, get_state(all) %.0f us, image() %.0f us (%.1f GB/s), 1000 ticks %.2f ms'
      % (N, N, up * 1e6, down * 1e6, img * 1e6, N * N * 4 / img / 1e9, run * 1e3))
print('  resident in HBM : %.0f Mcell-steps/s' % (cells / run / 1e6))
print('  upload + 1000 ticks + download: %.0f Mcell-steps/s' % (cells / (run + up + down) / 1e6))
print('  1000 ticks + 100 image() snapshots (the reference driver): %.0f Mcell-steps/s' % (cells / (run + 100 * img) / 1e6))
