"""Aliev-Panfilov two-variable model (Chaos, Solitons & Fractals 7:293, 1996), written the way the reference
writes its models: tuple state, ten chained solve() calls per tick."""
import numpy as np

import fib_tf_amd.tfgraph as tf
from fib_tf_amd.traced import IonicModel


class AlievPanfilov(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = 0.0
        self.max_v = 1.0
        self.depol = 0.0

    def rates(self, u, v):
        k, a, eps0, mu1, mu2 = 8.0, 0.15, 0.002, 0.2, 0.3
        eps = eps0 + mu1 * v / (u + mu2)
        du = -k * u * (u - a) * (u - 1) - u * v
        dv = eps * (-v - k * u * (u - a - 1))
        return du, dv

    def solve(self, state):
        u, v = state
        u0 = self.enforce_boundary(u)
        du, dv = self.rates(u, v)
        u1 = u0 + self.dt * du + self.diff * self.dt * self.laplace(u0)
        v1 = v + self.dt * dv
        return (u1, v1)

    def define(self, s1=True):
        super().define()
        u_init = np.zeros([self.height, self.width], dtype=np.float32)
        v_init = np.zeros([self.height, self.width], dtype=np.float32)
        if s1:
            u_init[:, :3] = 1.0
        u = tf.Variable(u_init, name='u')
        v = tf.Variable(v_init, name='v')
        states = [(u, v)]
        for i in range(10):
            states.append(self.solve(states[-1]))
        u1, v1 = states[-1]
        self.dt_per_step = 10
        self._ode_op = tf.group(u.assign(u1), v.assign(v1))
        self._u = u

    def pot(self):
        return self._u

    def image(self):
        return self._u.eval()
