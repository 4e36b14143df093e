"""The four-variable Cherry-Ehrlich-Nattel-Fenton atrial model written as a model file for the tracer: plain
TensorFlow-1 style (`tf.Variable`, `tf.sign`, `tf.tanh`, `tf.where`, `tf.assign`, `tf.group`), ten chained
sub-steps per tick.  Our own transcription of the published kinetics; it states the same float32 operation
sequence as the hand-written device code (fib_tf_amd/csrc/fenton_step.inc), so the kernel GENERATED from this
file must equal the hand-written kernel bit for bit under the rounding-faithful policy
(tests/test_gpu_traced.py::test_generated_four_variable_equals_handwritten)."""
import numpy as np
import tensorflow as tf
from ionic import IonicModel

# thresholds, time constants and shape parameters of the model
U_C, U_W, U_0, U_M, U_CSI, U_SO = 0.23, 0.146, 0.0, 1.0, 0.8, 0.3
TAU_D, TAU_SI, TAU_SO, TAU_A = 0.065, 31.8364, 31.8364, 0.009
A_SO, B_SO, C_SO = 0.115, 0.84, 0.02
TAU_VP, TAU_VN, TAU_WP, TAU_WN1, TAU_WN2 = 3.33, 19.2, 160.0, 75.0, 75.0
R_SP, R_SN, K_S = 0.02, 1.2, 3.0


def step_up(x):
    """0 below zero, 1 above, one half at zero"""
    return (1 + tf.sign(x)) * 0.5


def step_down(x):
    return (1 - tf.sign(x)) * 0.5


class FourVariable(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = 0.0
        self.max_v = 1.0
        self.depol = 0.0

    def rates(self, U, V, W, S):
        above_c = step_up(U - U_C)
        above_so = step_up(U - U_SO)
        below_so = step_down(U - U_SO)
        i_fast_in = -V * above_c * (U - U_C) * (U_M - U) / TAU_D
        i_slow_in = -W * S / TAU_SI
        i_slow_out = above_so * TAU_A + ((1 + tf.tanh((U - B_SO) / C_SO)) * (0.5 * (A_SO - TAU_A))
                                         + (U - U_0) * below_so / TAU_SO)
        dU = -(i_fast_in + i_slow_in + i_slow_out)
        excited = U > U_C
        dV = tf.where(excited, -V / TAU_VP, (1 - V) / TAU_VN)
        dW = tf.where(excited, -W / TAU_WP, tf.where(U > U_W, (1 - W) / TAU_WN2, (1 - W) / TAU_WN1))
        r_s = above_c * (R_SP - R_SN) + R_SN
        dS = r_s * ((1 + tf.tanh((U - U_CSI) * K_S)) * 0.5 - S)
        return dU, dV, dW, dS

    def solve(self, state):
        U, V, W, S = state
        U0 = self.enforce_boundary(U)
        dU, dV, dW, dS = self.rates(U, V, W, S)         # the reaction sees the raw potential
        U1 = U0 + self.dt * dU + self.diff * self.dt * self.laplace(U0)
        return U1, V + self.dt * dV, W + self.dt * dW, S + self.dt * dS

    def define(self, s1=True):
        super().define()
        shape = [self.height, self.width]
        u = np.zeros(shape, dtype=np.float32)
        if s1:
            u[:, 1] = 1.0
        vars_ = [tf.Variable(u, name='U'), tf.Variable(np.ones(shape, dtype=np.float32), name='V'),
                 tf.Variable(np.ones(shape, dtype=np.float32), name='W'),
                 tf.Variable(np.zeros(shape, dtype=np.float32), name='S')]
        chain = [tuple(vars_)]
        for _ in range(10):
            chain.append(self.solve(chain[-1]))
        self.dt_per_step = 10
        self._ode_op = tf.group(*[tf.assign(v, new) for v, new in zip(vars_, chain[-1])])
        self._U = vars_[0]

    def pot(self):
        return self._U

    def image(self):
        return self._U.eval()
