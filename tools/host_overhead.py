#!/usr/bin/env python3
"""GPU box: host-side cost of enqueueing ticks (tiny grid, so the GPU is never the bottleneck)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd import _lib
for model, name in ((0, 'fenton'), (1, 'br'), (2, 'court')):
    for k in ('', '1'):
        if k: os.environ['FIBHIP_K'] = k
        else: os.environ.pop('FIBHIP_K', None)
        st = _lib.Stepper(model, 32, 32, 0.1, 1.0, flags=_lib.FAST)
        fused, per_tick = st.launch_plan()
        st.step(50); st.sync()
        n = 2000
        t0 = time.perf_counter(); st.step(n); t1 = time.perf_counter(); st.sync(); t2 = time.perf_counter()
        t3 = time.perf_counter()
        for _ in range(n): st.step(1)
        t4 = time.perf_counter(); st.sync(); t5 = time.perf_counter()
        print('%-7s K=%-2d launches/tick %2d: C loop enqueue %.2f us/tick (%.2f us/launch), drained +%.2f us/tick | python loop %.2f us/tick, drained +%.2f'
              % (name, fused, per_tick, (t1 - t0) / n * 1e6, (t1 - t0) / n / per_tick * 1e6, (t2 - t1) / n * 1e6, (t4 - t3) / n * 1e6, (t5 - t4) / n * 1e6))
        st.close()
