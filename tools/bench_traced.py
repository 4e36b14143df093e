"""time the kernels GENERATED from model files (tests/models/four_variable.py, eight_variable.py: our own
transcriptions of the two published models) next to the hand-written kernels: python tools/bench_traced.py [size]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fib_tf_amd import _lib  # noqa: E402
from traced_cases import make_model  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def phase(n):
    yy, xx = np.mgrid[0:n, 0:n]
    return np.maximum(0.5 * (np.tanh(np.hypot(xx - n / 2, yy - n / 2) - n / 17) + 1), 1e-5).astype(np.float32)


def rate(st, ticks):
    st.step(20)
    st.sync()
    best = min(st.time_steps(ticks)[0] for _ in range(3))
    return N * N * ticks * st.steps_per_tick / (best * 1e-3) / 1e6, best / ticks * 1e3


def init_for(model):
    if model == 'fenton':
        s = np.zeros((4, N, N), np.float32); s[1:3] = 1; s[0][:, 1] = 1
    else:
        s = np.empty((8, N, N), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            s[i] = v
        s[0][:, 1] = 10
    return s


for name, model, mid, ticks in (('fv', 'fenton', _lib.FENTON4V, 500), ('ev', 'br', _lib.BR, 300)):
    ph, init = phase(N), init_for(model)
    for fast in (True, False):
        m = make_model(name, N, N, fast_math=fast)
        m.phase = ph
        m.define()
        m._ensure_compiled()                # (the generated kernels are built on first use)
        m._stepper.set_state(-1, init)
        tr, tus = rate(m._stepper, ticks)
        tp = m._stepper.launch_plan()
        m._stepper.close()
        st = _lib.Stepper(mid, N, N, m.dt, m.diff, flags=_lib.FAST if fast else 0)
        st.set_phase(ph)
        st.set_state(-1, init)
        nr, nus = rate(st, ticks)
        np_ = st.launch_plan()
        st.close()
        print('%-4s %4d^2 %-5s generated %9.0f Mcs/s (%7.2f us/tick, plan %s)   hand-written %9.0f Mcs/s (%7.2f us/tick, plan %s)   ratio %.2f'
              % (name, N, 'fast' if fast else 'exact', tr, tus, tp, nr, nus, np_, tr / nr))
