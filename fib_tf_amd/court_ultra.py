"""court_ultra — the single-rate research variant of the Courtemanche model
(siravan/fib_tf `court_ultra.py:32-559`): every tick assigns ALL state variables with dt
(`court_ultra.py:107-111,127-128`) instead of court.py's fast/slow split, `fire_op('slow')` is an empty
op there (`court_ultra.py:110`), and run states are checkpointed with `np.save(name, m.state)` /
`np.load(name).item(0)` (`court_ultra.py:511,518`).

Device side: the same kernel as court.py instantiated with `MODE_ALL` (flag FIBHIP_ALLVARS).
`config['ultra_slow'] = True` (the optional 22nd `_us_` gate, `court_ultra.py:81-82,198-199,221-222,445-450`)
is not implemented; the reference's own driver runs with it off (`court_ultra.py:543`)."""
import numpy as np

from . import _lib
from .court import Courtemanche as _Courtemanche


class Courtemanche(_Courtemanche):
    def __init__(self, props):
        super().__init__(props)
        if getattr(self, 'ultra_slow', False):
            raise NotImplementedError("court_ultra: config['ultra_slow']=True (the _us_ gate) is not implemented")
        self.ultra_slow = False

    def _flags(self):
        return super()._flags() | _lib.ALLVARS

    def define(self, s1=True, state=None):
        super().define(s1=s1, state=state)
        self._ops['slow'] = ('call', lambda: None)             # tf.group() of nothing, court_ultra.py:110

    def _fire_trend(self):
        # only V, at [width//2, height//8] (court_ultra.py:112-116)
        v = self._stepper.probe(0, self.width // 2, self.height // 8)
        self._Trend.value = np.array([v, 0.0], dtype=np.float32)

    def solve(self, State):
        """one single-rate evaluation of all 21 variables on host arrays (court_ultra.py:134-262)"""
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
