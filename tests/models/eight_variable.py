"""The Beeler-Reuter ventricular model (1977; calcium gates sped up twofold) written as a model file for the
tracer: eight state arrays, Rush-Larsen gates from the published rate table, five chained sub-steps per tick.
Our own transcription; the kernel GENERATED from this file is compared with the hand-written Beeler-Reuter
kernel (direct gates) on the GPU, to rounding (tests/test_gpu_traced.py::test_generated_eight_variable_vs_handwritten)."""
import numpy as np
import tensorflow as tf
from ionic import IonicModel

# rate = (a * exp(b * (v + c)) + d * (v + e)) / (exp(f * (v + c)) + g), Beeler & Reuter 1977, table 1;
# name -> ((alpha row), (beta row)).  The d and f gates run at twice the published rates.
RATES = {
    'x1': ((0.0005, 0.083, 50.0, 0.0, 0.0, 0.057, 1.0), (0.0013, -0.06, 20.0, 0.0, 0.0, -0.04, 1.0)),
    'm': ((0.0, 0.0, 47.0, -1.0, 47.0, -0.1, -1.0), (40.0, -0.056, 72.0, 0.0, 0.0, 0.0, 0.0)),
    'h': ((0.126, -0.25, 77.0, 0.0, 0.0, 0.0, 0.0), (1.7, 0.0, 22.5, 0.0, 0.0, -0.082, 1.0)),
    'j': ((0.055, -0.25, 78.0, 0.0, 0.0, -0.2, 1.0), (0.3, 0.0, 32.0, 0.0, 0.0, -0.1, 1.0)),
    'd': ((2 * 0.095, -0.01, -5.0, 0.0, 0.0, -0.072, 1.0), (2 * 0.07, -0.017, 44.0, 0.0, 0.0, 0.05, 1.0)),
    'f': ((2 * 0.012, -0.008, 28.0, 0.0, 0.0, 0.15, 1.0), (2 * 0.0065, -0.02, 30.0, 0.0, 0.0, -0.2, 1.0)),
}


def rate(v, row):
    a, b, c, d, e, f, g = [float(np.float32(x)) for x in row]
    top = a * tf.exp(b * (v + c))
    if d != 0.0:
        top = top + d * (v + e)
    return top / (tf.exp(f * (v + c)) + g)


class EightVariable(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = -90.0
        self.max_v = 30.0
        self.depol = -84.6

    def gate(self, g, name, v):
        alpha, beta = rate(v, RATES[name][0]), rate(v, RATES[name][1])
        return self.rush_larsen(g, alpha / (alpha + beta), 1.0 / (alpha + beta), self.dt)

    def solve(self, state):
        V, C, M, H, J, D, F, X = state
        V0 = self.enforce_boundary(V)
        M1, H1, J1 = self.gate(M, 'm', V0), self.gate(H, 'h', V0), self.gate(J, 'j', V0)
        D1, F1, X1 = self.gate(D, 'd', V0), self.gate(F, 'f', V0), self.gate(X, 'x1', V0)
        # time-independent and time-activated outward potassium, fast sodium, slow inward calcium — from the OLD gates
        i_k1 = 0.35 * (4.0 * (tf.exp(0.04 * (V0 + 85.0)) - 1.0) / (tf.exp(0.08 * (V0 + 53.0)) + tf.exp(0.04 * (V0 + 53.0)))
                       + 0.2 * ((V0 + 23.0) / (1.0 - tf.exp(-0.04 * (V0 + 23.0)))))
        i_x1 = X * 0.8 * (tf.exp(0.04 * (V0 + 77.0)) - 1.0) / tf.exp(0.04 * (V0 + 35.0))
        i_na = 1.0 * (4.0 * M * M * M * H * J + 0.005) * (V0 - 50.0)
        e_ca = -82.3 - 13.0278 * tf.log(C)
        i_ca = 0.09 * D * F * (V0 - e_ca)
        total = i_k1 + i_x1 + i_na + i_ca
        V1 = tf.clip_by_value(V0 + self.diff * self.dt * self.laplace(V0) - self.dt * total, -85.0, 25.0)
        C1 = C + self.dt * (-1.0e-7 * i_ca + 0.07 * (1.0e-7 - C))
        return V1, C1, M1, H1, J1, D1, F1, X1

    def define(self, s1=True):
        super().define()
        shape = [self.height, self.width]
        rest = (('V', -84.624), ('C', 1e-4), ('M', 0.01), ('H', 0.988), ('J', 0.975), ('D', 0.003), ('F', 0.994),
                ('XI', 0.0001))
        init = {n: np.full(shape, x, dtype=np.float32) for n, x in rest}
        if s1:
            init['V'][:, 1] = 10.0
        vars_ = [tf.Variable(init[n], name=n) for n, _ in rest]
        chain = [tuple(vars_)]
        for _ in range(5):
            chain.append(self.solve(chain[-1]))
        self.dt_per_step = 5
        self._ode_op = tf.group(*[tf.assign(v, new) for v, new in zip(vars_, chain[-1])])
        self._V = vars_[0]

    def pot(self):
        return self._V

    def image(self):
        return (self._V.eval() - self.min_v) / (self.max_v - self.min_v)
