#!/usr/bin/env python3
"""Where do the microseconds of a 20-tick benchmark region go that are not kernel time?  (round-3 review item 3)
One region = expect(20), 20 x step(1), sync() — bench.py's `--steps 20` region.  Host clock around the three phases, HIP
events (time_begin / time_end on the kernel's own stream) around the same series in alternating repetitions."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd.fenton import Fenton4v

m = Fenton4v({'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 1.5, 'duration': 1000, 'skip': False, 'cheby': True})
m.add_hole_to_phase_field(256, 256, 30)
m.define()
st = m._stepper
st.step(1)
st.sync()
for _ in range(200):                       # clocks up
    st.expect(20)
    for _ in range(20):
        st.step(1)
    st.sync()
import gc
gc.collect()
gc.freeze()
first, rest, wait, total, ev = [], [], [], [], []
N = 20
for rep in range(300):
    if rep % 2 == 0:
        st.sync()
        t0 = time.perf_counter()
        st.expect(N)
        st.step(1)
        t1 = time.perf_counter()
        for _ in range(N - 1):
            st.step(1)
        t2 = time.perf_counter()
        st.sync()
        t3 = time.perf_counter()
        first.append((t1 - t0) * 1e6)
        rest.append((t2 - t1) * 1e6)
        wait.append((t3 - t2) * 1e6)
        total.append((t3 - t0) * 1e6)
    else:
        st.sync()
        st.time_begin()
        st.expect(N)
        for _ in range(N):
            st.step(1)
        ms, launches = st.time_end()
        ev.append(ms * 1e3)
med = statistics.median
print('one region of %d ticks, 512x512 Fenton, medians of %d repetitions (us):' % (N, len(total)))
print('  host clock: expect + first step() (the launch call) %.1f | the other %d step() calls %.1f | sync() %.1f | whole region %.1f'
      % (med(first), N - 1, med(rest), med(wait), med(total)))
print('  HIP events around the same series on the stream: %.1f (one launch: prologue + %d ticks + write-back) = %.2f per tick' % (med(ev), N, med(ev) / N))
print('  => outside the events: %.1f us (the launch reaching the device + the polling host noticing the end)' % (med(total) - med(ev)))
st.time_begin()
st.step(32 * 16)
ms, launches = st.time_end()
print('  for comparison, 16 launches of 32 ticks back to back: %.2f us per tick' % (ms * 1e3 / (32 * 16)))
