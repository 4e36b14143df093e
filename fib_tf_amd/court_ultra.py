"""court_ultra — the single-rate research variant of the Courtemanche model
(siravan/fib_tf `court_ultra.py:32-559`): every tick assigns ALL state variables with dt
(`court_ultra.py:107-111,127-128`) instead of court.py's fast/slow split, `fire_op('slow')` is an empty
op there (`court_ultra.py:110`), and run states are checkpointed with `np.save(name, m.state)` /
`np.load(name).item(0)` (`court_ultra.py:511,518`).

Device side: the same kernel as court.py instantiated with `MODE_ALL` (flag FIBHIP_ALLVARS).
`config['ultra_slow'] = True` adds the 22nd array `_us_`, an ultra-slow gate that scales i_Na
(`court_ultra.py:81-82,198-199,221-222,445-450`): device model `CourtemancheUS` (FIBHIP_COURT_US)."""
from functools import partial

import numpy as np

from . import _lib
from .court import INITIAL, Courtemanche as _Courtemanche


class _InterVar:
    """`m._Inter[key].eval()` (court_ultra.py:113,475-479): the intermediate evaluated on the current V"""

    def __init__(self, model, key):
        self._model, self._key = model, key

    def eval(self):
        return self._model.calc_inter(self._model._V.eval())[self._key]


class Courtemanche(_Courtemanche):
    def __init__(self, props):
        super().__init__(props)
        self.ultra_slow = bool(getattr(self, 'ultra_slow', False))
        if self.ultra_slow:                                    # instance-level: the 22-array model
            self.MODEL_ID = _lib.COURT_US
            self.VAR_NAMES = tuple(n for n, _ in INITIAL) + ('_us_',)

    def _flags(self):
        return super()._flags() | _lib.ALLVARS

    def define(self, s1=True, state=None):
        if state is None and self.ultra_slow:
            state = {}
            for name, value in INITIAL:
                self.init_state_variable(state, name, value)
            self.init_state_variable(state, '_us_', 0.72)      # steady state at 500 ms, court_ultra.py:81-82
            if s1:
                state['V'][:, :25] = 20.0
        super().define(s1=s1, state=state)
        self._ops['slow'] = ('call', lambda: None)             # tf.group() of nothing, court_ultra.py:110
        self._Inter = {k: _InterVar(self, k) for k in _lib.COURT_INTER_KEYS}

    def _fire_trend(self):
        # only V, at [width//2, height//8] (court_ultra.py:112-116)
        v = self._stepper.probe(0, self.width // 2, self.height // 8)
        self._Trend.value = np.array([v, 0.0], dtype=np.float32)

    def solve(self, State):
        """one single-rate evaluation of all variables on host arrays (court_ultra.py:134-262)"""
        arrs = np.stack([np.asarray(State[n], np.float32) for n in self.VAR_NAMES])
        st = self._new_stepper(steps_per_tick=1, shard=False)
        try:
            st.set_state(-1, arrs)
            st.step(1)
            res = st.get_state(-1)
        finally:
            st.close()
        return {n: res[i] for i, n in enumerate(self.VAR_NAMES)}


def save_state(path, state):
    """`np.save('state_small', m.state)` of the reference (court_ultra.py:511): a pickled dict name -> array"""
    np.save(path, state, allow_pickle=True)


def load_state(path):
    """`np.load('state_small.npy').item(0)` of the reference (court_ultra.py:518)"""
    if not str(path).endswith('.npy'):
        path = str(path) + '.npy'
    return np.load(path, allow_pickle=True).item(0)


def cl_observer(m, cyclelengths, i0, i, cl):
    """the ϕ-weighted cycle-length observer of court_ultra.py:465-486 (bind with functools.partial)"""
    mean_na = np.average(m._State['_Na_i_'].eval(), weights=m.phase)
    mean_ca = np.average(m._State['_f_Ca_'].eval(), weights=m.phase)
    if m.ultra_slow:
        mean_us = np.average(m._State['_us_'].eval(), weights=m.phase)
        inter = m.calc_inter(m._V.eval())
        mean_us_infinity = np.average(inter['us_infinity'], weights=m.phase)
        mean_tau_us = np.average(inter['tau_us'], weights=m.phase)
        cyclelengths.append([i0 + i, cl, mean_na, mean_ca, mean_us, mean_us_infinity, mean_tau_us])
        print('%d:\t%d\t%.3f\t%.3f\t%.5f\t%.5f\t%.0f' % (i + i0, cl, mean_na, mean_ca, mean_us, mean_us_infinity,
                                                       mean_tau_us))
    else:
        cyclelengths.append([i0 + i, cl, mean_na, mean_ca])
        print('%d:\t%d\t%.3f\t%.3f' % (i + i0, cl, mean_na, mean_ca))


def run_small(config, im, cyclelengths, radius=50, i0=0, state_file='state_small'):
    """the two-stage protocol's first stage (court_ultra.py:489-512): annular domain, S1-S2, checkpoint"""
    m = Courtemanche(config)
    m.add_hole_to_phase_field(m.width // 2, m.height // 2, radius)
    m.add_hole_to_phase_field(m.width // 2, m.height // 2, m.width // 2 - 6, neg=True)
    m.define()
    m.add_pace_op('s2', 'luq', 10.0)
    m.cl_observer = partial(cl_observer, m, cyclelengths, i0)
    s2 = m.millisecond_to_step(300)
    for i in m.run(im, keep_state=True, block=False):
        if i % 10 == 0:
            m.fire_op('slow')
        if i == s2:
            m.fire_op('s2')
        if i % 5000 == 0:
            image, phase = m.image(), m.phase
            rho = np.sum(image[phase > 1e-3] < 0.2) / np.sum(phase > 1e-3)     # cutoff -55 mV
            print('ρ = %.4f' % rho)
    save_state(state_file, m.state)
    return m.state


def run_large(config, im, cyclelengths, radius, i0=0, state_file='state_small', out_file='state_large'):
    """second stage (court_ultra.py:514-528): resume the checkpoint around a different obstacle"""
    m = Courtemanche(config)
    m.add_hole_to_phase_field(m.width // 2, m.height // 2, radius)
    m.define(state=load_state(state_file))
    m.cl_observer = partial(cl_observer, m, cyclelengths, i0)
    for i in m.run(im, keep_state=True, block=False):
        if i % 10 == 0:
            m.fire_op('slow')
    save_state(out_file, m.state)
    return m.state
