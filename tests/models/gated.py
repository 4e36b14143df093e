"""A three-current excitable-membrane model in the style of the reference's court.py: dict state, voltage-only
intermediates, Rush-Larsen gates, a fast/slow split of the assign ops ('slow' fired by the driver every 10th
tick with 10*dt), a Trend probe, one solve() per tick.  The file imports `tensorflow` and `ionic` exactly as a
reference model file does — it runs after fib_tf_amd.tfgraph.install()."""
import numpy as np
import tensorflow as tf
from ionic import IonicModel


class Gated(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = -90.0
        self.max_v = 40.0
        self.depol = -80.0
        self.fast_states = ['V', 'm']

    def δt(self, name):
        return self.dt if name in self.fast_states else self.dt * 10

    def calc_inter(self, V):
        inter = {}
        inter['m_inf'] = tf.reciprocal(1.0 + tf.exp((V + 35.0) / -7.0))
        inter['tau_m'] = 0.05 + 0.3 * tf.exp(-tf.square((V + 40.0) / 30.0))
        inter['h_inf'] = 0.5 * (1 - tf.tanh((V + 62.0) / 12.0))
        inter['tau_h'] = 2.0 + 18.0 / (1.0 + tf.exp((V + 50.0) / 8.0))
        a_n = tf.where(tf.abs(V + 20.0) < 1e-6, 0.1 + V * 0.0, 0.01 * (V + 20.0) / (1.0 - tf.exp(-(V + 20.0) / 10.0)))
        b_n = 0.125 * tf.exp(-(V + 30.0) / 80.0)
        inter['n_inf'] = a_n / (a_n + b_n)
        inter['tau_n'] = 4.0 * tf.reciprocal(a_n + b_n)
        return inter

    def solve(self, State):
        V = self.enforce_boundary(State['V'])
        inter = self.calc_inter(V)
        g_Na, g_K, g_L, E_Na, E_K, E_L, Cm = 12.0, 3.6, 0.08, 50.0, -85.0, -70.0, 1.0
        State1 = {}
        State1['m'] = self.rush_larsen(State['m'], inter['m_inf'], inter['tau_m'], self.δt('m'))
        State1['h'] = self.rush_larsen(State['h'], inter['h_inf'], inter['tau_h'], self.δt('h'))
        State1['n'] = self.rush_larsen(State['n'], inter['n_inf'], inter['tau_n'], self.δt('n'))
        i_Na = g_Na * tf.pow(State['m'], 3) * State['h'] * (V - E_Na)
        i_K = g_K * tf.sqrt(tf.maximum(State['n'], 1e-6)) * State['n'] * (V - E_K)
        i_L = g_L * (V - E_L)
        i_ion = i_Na + i_K + i_L
        State1['V'] = V + self.dt * (-i_ion / Cm) + self.diff * self.dt * self.laplace(V)
        # a slow concentration-like variable driven by the sodium current
        State1['c'] = State['c'] + self.δt('c') * (-1e-4 * i_Na - 0.02 * (State['c'] - 1.0))
        return State1, inter

    def define(self, s1=True):
        super().define()
        state = {}
        for name, value in (('V', -80.0), ('m', 0.002), ('h', 0.95), ('n', 0.02), ('c', 1.0)):
            state[name] = np.full([self.height, self.width], value, dtype=np.float32)
        if s1:
            state['V'][:, :4] = 10.0
        State = {}
        for s in state:
            State[s] = tf.Variable(state[s])
        State1, inter = self.solve(State)
        self.dt_per_step = 1
        fasts, slows = [], []
        for s in State:
            (fasts if s in self.fast_states else slows).append(tf.assign(State[s], State1[s]))
        self._ode_op = tf.group(*fasts)
        self._ops['slow'] = tf.group(*slows)
        self._V = State['V']
        self._State = State
        Trend = tf.Variable(np.zeros([2], dtype=np.float32))
        self._ops['trend'] = tf.group(tf.assign(Trend[0], self._V[self.height // 2, 6]),
                                      tf.assign(Trend[1], State['c'][self.height // 2, 6]))
        self._Trend = Trend

    def pot(self):
        return self._V

    def image(self):
        v = self._V.eval()
        return (v - self.min_v) / (self.max_v - self.min_v)
