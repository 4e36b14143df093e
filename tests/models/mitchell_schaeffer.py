"""Mitchell-Schaeffer two-current model (Bull. Math. Biol. 65:767, 2003): list state, a gate switched by
tf.where, five chained solve() calls per tick, tf.assign / tf.square spellings."""
import numpy as np

import fib_tf_amd.tfgraph as tf
from fib_tf_amd.traced import IonicModel


class MitchellSchaeffer(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = 0.0
        self.max_v = 1.0
        self.depol = 0.0

    def solve(self, state):
        v, h = state
        tau_in, tau_out, tau_open, tau_close, v_gate = 0.3, 6.0, 120.0, 150.0, 0.13
        v0 = self.enforce_boundary(v)
        j_in = h * tf.square(v) * (1.0 - v) / tau_in
        j_out = -v / tau_out
        dh = tf.where(v < v_gate, (1.0 - h) / tau_open, -h / tau_close)
        v1 = tf.clip_by_value(v0 + self.dt * (j_in + j_out) + self.diff * self.dt * self.laplace(v0), 0.0, 1.0)
        h1 = h + self.dt * dh
        return [v1, h1]

    def define(self, s1=True):
        super().define()
        v_init = np.zeros([self.height, self.width], dtype=np.float32)
        h_init = np.ones([self.height, self.width], dtype=np.float32)
        if s1:
            v_init[:4, :] = 0.9
        V = tf.Variable(v_init, name='v')
        Hg = tf.Variable(h_init, name='h')
        state = [V, Hg]
        for i in range(5):
            state = self.solve(state)
        self.dt_per_step = 5
        self._ode_op = tf.group(tf.assign(V, state[0]), tf.assign(Hg, state[1]))
        self._V = V
        self._State = {'v': V, 'h': Hg}

    def pot(self):
        return self._V

    def image(self):
        return self._V.eval()
