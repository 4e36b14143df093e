"""debug: where does a T-tick launch differ from T launches of one tick?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd import _lib


def run(H, W, T, mt, phase, variant, fast=True):
    os.environ['FIBHIP_MT'] = '1' if mt else '0'
    os.environ['FIBHIP_VARIANT'] = variant
    rng = np.random.default_rng(1)
    init = np.empty((4, H, W), np.float32)
    init[0] = rng.uniform(-0.02, 1.0, (H, W))
    for v in (1, 2, 3):
        init[v] = rng.uniform(0, 1, (H, W))
    phi = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
    st = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, 1.3, flags=_lib.FAST if fast else 0)
    if phase:
        st.set_phase(phi)
    st.set_state(-1, init)
    st.step(T)
    out = st.get_state(-1)
    tpl = st.ticks_per_launch()
    st.close()
    return out, tpl


for (H, W, variant) in [(20, 40, '10,44,25,-3'), (20, 100, '10,44,25,-3'), (60, 40, '10,44,25,-3'), (64, 64, '10,44,25,-3'), (512, 512, '10,44,25,-3')]:
    for T in (2, 3):
        for phase in (False, True):
            a, tpl = run(H, W, T, True, phase, variant)
            b, _ = run(H, W, T, False, phase, variant)
            d = (a != b)
            msg = '%dx%d T=%d phase=%d tpl=%d: ' % (H, W, T, phase, tpl)
            if not d.any():
                print(msg + 'identical')
                continue
            for v in range(4):
                ys, xs = np.nonzero(d[v])
                if len(ys):
                    print(msg + 'var %d: %d cells differ, rows %d..%d cols %d..%d, max |d| %.3g; nan %d' % (
                        v, len(ys), ys.min(), ys.max(), xs.min(), xs.max(), float(np.nanmax(np.abs(a[v] - b[v]))), int(np.isnan(a[v]).sum())))
            if H <= 64 and W <= 64:
                m = d[0]
                for y in range(H):
                    print(''.join('x' if m[y, x] else '.' for x in range(W)))
