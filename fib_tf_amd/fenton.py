"""Fenton4v — the Cherry-Ehrlich-Nattel-Fenton 4-variable canine left-atrial model behind the
reference's API (siravan/fib_tf `fenton.py:31-153`).  The kinetics (fenton.py:46-92) and the
explicit-Euler update (fenton.py:95-108) live in the HIP kernel (csrc/models.hpp `Fenton`);
this class only owns initial conditions, the handle and the read-back."""
import numpy as np

from . import _lib
from .ionic import IonicModel


class Fenton4v(IonicModel):
    MODEL_ID = _lib.FENTON4V
    VAR_NAMES = ('U', 'V', 'W', 'S')

    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = 0.0            # fenton.py:42-44
        self.max_v = 1.0
        self.depol = 0.0

    def define(self, s1=True, state=None):
        """initial conditions u=0, v=1, w=1, s=0 and the S1 stimulus column u[:,1]=1
        (fenton.py:116-123); one tick = 10 fused sub-steps (fenton.py:133-138).
        state: resume from a dict U/V/W/S -> [H,W] array instead (what `run(keep_state=True)` leaves in
        `model.state`; the reference offers this for Courtemanche only, court.py:87-89 — same contract here)"""
        super().define()
        if state is not None:
            self._create(self._resume_arrays(state))
            self._U = self._State['U']
            return
        shape = [self.height, self.width]
        u_init = np.zeros(shape, dtype=np.float32)
        v_init = np.ones(shape, dtype=np.float32)
        w_init = np.ones(shape, dtype=np.float32)
        s_init = np.zeros(shape, dtype=np.float32)
        if s1:
            u_init[:, 1] = 1.0
        self._create([u_init, v_init, w_init, s_init])
        self._U = self._State['U']

    def solve(self, state):
        """ONE explicit-Euler sub-step of (U, V, W, S) host arrays on the GPU (fenton.py:95-108)"""
        st = self._new_stepper(steps_per_tick=1, shard=False)
        try:
            st.set_state(-1, np.stack([np.asarray(a, np.float32) for a in state]))
            st.step(1)
            return tuple(st.get_state(-1))
        finally:
            st.close()

    def pot(self):
        return self._U

    def image(self):
        return self._U.eval()
