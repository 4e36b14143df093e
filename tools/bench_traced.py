"""time the libraries generated from the reference's unchanged model files (oracle/_ref/traced/*.so) next to the
hand-written kernels: python tools/bench_traced.py [size]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fib_tf_amd import _lib  # noqa: E402
from fib_tf_amd.br import BeelerReuter  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
TR = os.path.join(ROOT, 'oracle', '_ref', 'traced')


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
    elif model == 'br':
        s = np.empty((8, N, N), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            s[i] = v
        s[0][:, 1] = 10
    else:
        from fib_tf_amd.court import INITIAL
        s = np.empty((21, N, N), np.float32)
        for i, (_, v) in enumerate(INITIAL):
            s[i] = v
        s[0][:, :25] = 20
    return s


for case, model, mid, flags, ticks in (('fenton_d1.5', 'fenton', _lib.FENTON4V, 0, 500),
                                       ('br_cheby_d0.809', 'br', _lib.BR, _lib.CHEBY, 300),
                                       ('court_d0.809', 'court', _lib.COURT, _lib.CHRONIC, 500)):
    so = os.path.join(TR, case + '.so')
    if not os.path.exists(so):
        print(case, 'not built')
        continue
    meta = json.load(open(os.path.join(TR, case + '.json')))
    ph, init = phase(N), init_for(model)
    for fast in (1, 0):
        row = []
        hand = None
        if model == 'br':                                   # the product path: table-specialised build (fib_tf_amd/br.py)
            from fib_tf_amd.br import specialised_library
            hand = specialised_library(BeelerReuter({'height': 8, 'width': 8, 'cheby': True})._table32())
        for lib, m, fl in ((_lib.load(so), _lib.CUSTOM, 0), (hand, mid, flags)):
            st = _lib.Stepper(m, N, N, meta['dt'], meta['diff'], flags=fl | (_lib.FAST if fast else 0), library=lib)
            if model == 'br' and m == mid:
                st.set_consts(BeelerReuter({'height': 8, 'width': 8}).chebyshev_table())
            st.set_phase(ph)
            st.set_state(-1, init)
            row.append(rate(st, ticks) + (st.launch_plan(),))
            st.close()
        (tr, tus, tp), (nr, nus, np_) = row
        print('%-18s %4d^2 %-5s traced %9.0f Mcs/s (%7.2f us/tick, plan %s)   hand-written %9.0f Mcs/s (%7.2f us/tick, plan %s)   ratio %.2f'
              % (case, N, 'fast' if fast else 'exact', tr, tus, tp, nr, nus, np_, tr / nr))
