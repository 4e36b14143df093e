"""Courtemanche — the modified 21-variable Courtemanche human atrial model behind the reference's
API (siravan/fib_tf `court.py:31-580`).  Device side: csrc/models.hpp `Courtemanche`.

Multirate scheme as in the reference: every tick assigns the fast set {V, Na_i, m, h} with dt
(court.py:42,94-102); the driver fires 'slow' every 10th tick, which re-evaluates the model on the
post-fast state and assigns the other 17 variables with 10·dt (court.py:103,118-122,615-617)."""
import numpy as np

from . import _lib
from .ionic import IonicModel

# insertion order of court.py:57-78 == variable index in the slab
INITIAL = (('V', -81.18), ('_Na_i_', 1.117e+01), ('_m_', 2.98e-3), ('_h_', 9.649e-1), ('_j_', 9.775e-1),
           ('_K_i_', 1.39e+02), ('_oa_', 3.043e-2), ('_oi_', 9.992e-1), ('_ua_', 4.966e-3), ('_ui_', 9.986e-1),
           ('_xr_', 3.296e-5), ('_xs_', 1.869e-2), ('_Ca_i_', 1.013e-4), ('_d_', 1.367e-4), ('_f_', 9.996e-1),
           ('_f_Ca_', 7.755e-1), ('_Ca_rel_', 1.488), ('_u_', 0.0), ('_v_', 1.0), ('_w_', 0.9992),
           ('_Ca_up_', 1.488))


class _Trend:
    """the 2-element probe variable of court.py:107-112"""

    def __init__(self):
        self.value = np.zeros([2], dtype=np.float32)

    def eval(self):
        return self.value.copy()


class Courtemanche(IonicModel):
    MODEL_ID = _lib.COURT
    VAR_NAMES = tuple(n for n, _ in INITIAL)

    def __init__(self, props):
        super().__init__(props)
        self.min_v = -100.0         # mV, court.py:38-42
        self.max_v = 50.0
        self.depol = -81.0
        self.chronic = True
        self.fast_states = ['V', '_Na_i_', '_m_', '_h_']

    def _flags(self):
        return super()._flags() | (_lib.CHRONIC if self.chronic else 0)

    def init_state_variable(self, state, name, value):
        if name in state:
            print('Warning! The state variable arlready exists')
        state[name] = np.full([self.height, self.width], value, dtype=np.float32)

    def define(self, s1=True, state=None):
        """initial conditions court.py:57-82 (S1: V[:, :25] = 20 mV) or resume from `state`
        (a dict name -> [H,W] array, e.g. a previous run's `model.state`, court.py:623-626)"""
        IonicModel.define(self)
        if state is None:
            state = {}
            for name, value in INITIAL:
                self.init_state_variable(state, name, value)
            if s1:
                state['V'][:, :25] = 20.0
        missing = [n for n in self.VAR_NAMES if n not in state]
        if missing:
            raise KeyError('define(state=...): missing state variables %s' % missing)
        self._create([state[n] for n in self.VAR_NAMES])
        self._V = self._State['V']
        self._ops['slow'] = ('call', self._stepper.step_slow)
        self._Trend = _Trend()
        self._ops['trend'] = ('call', self._fire_trend)

    def _fire_trend(self):
        # V and Na_i at [width//2, 20], court.py:107-111
        r, c = self.width // 2, 20
        st = self._stepper
        self._Trend.value = np.array([st.probe(0, r, c), st.probe(1, r, c)], dtype=np.float32)

    def solve(self, State):
        """ONE evaluation of the model on a dict of host arrays: returns all 21 new arrays, the
        fast set advanced by dt and the slow set by 10·dt, as `solve` of court.py:124-271 does"""
        arrs = np.stack([np.asarray(State[n], np.float32) for n in self.VAR_NAMES])
        out = {}
        for which in ('fast', 'slow'):
            st = self._new_stepper(steps_per_tick=1, shard=False)
            try:
                st.set_state(-1, arrs)
                st.step(1) if which == 'fast' else st.step_slow()
                res = st.get_state(-1)
            finally:
                st.close()
            for i, n in enumerate(self.VAR_NAMES):
                if (n in self.fast_states) == (which == 'fast'):
                    out[n] = res[i]
        return out

    def calc_inter(self, V, mod=None):
        """the voltage-only intermediates (court.py:273-429; court_ultra.py:264-452 adds us_infinity, tau_us), evaluated by the device code;
        `mod` (np / tf in the reference) is accepted and ignored"""
        return _lib.court_inter(V, fast=bool(getattr(self, 'fast_math', True)), device=self.device)

    def pot(self):
        return self._V

    def image(self):
        """V scaled to 0..1 (court.py:574-580)"""
        v = self._V.eval()
        return (v - self.min_v) / (self.max_v - self.min_v)


def cl_observer(i, cl):
    print('Observer: %d:\t%d' % (i, cl))
