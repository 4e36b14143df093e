"""FitzHugh-Nagumo with a recovery variable advanced on a coarser clock, in the style of the reference's
br.py `skip` option: solve(state, n) advances w by n*dt; a tick is solve(., 4) followed by three solve(., 0)."""
import numpy as np
import tensorflow as tf
from ionic import IonicModel


class MultirateFHN(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = -2.5
        self.max_v = 2.5
        self.depol = -1.2

    def solve(self, state, n):
        v, w = state
        v0 = self.enforce_boundary(v)
        dv = 3.0 * (v - tf.pow(v, 3.0) / 3.0 - w)
        v1 = v0 + self.dt * dv + self.diff * self.dt * self.laplace(v0)
        if n > 0:
            w1 = w + (n * self.dt) * (0.08 * (v + 0.7 - 0.8 * w))
        else:
            w1 = w
        return v1, w1

    def define(self, s1=True):
        super().define()
        v_init = np.full([self.height, self.width], -1.2, dtype=np.float32)
        w_init = np.full([self.height, self.width], -0.62, dtype=np.float32)
        if s1:
            v_init[:, :4] = 1.5
        v = tf.Variable(v_init, name='v')
        w = tf.Variable(w_init, name='w')
        states = [(v, w)]
        states.append(self.solve(states[-1], 4))
        for i in range(3):
            states.append(self.solve(states[-1], 0))
        v1, w1 = states[-1]
        self.dt_per_step = 4
        self._ode_op = tf.group(tf.assign(v, v1, name='set_v'), tf.assign(w, w1, name='set_w'))
        self._v = v

    def pot(self):
        return self._v

    def image(self):
        return (self._v.eval() - self.min_v) / (self.max_v - self.min_v)
