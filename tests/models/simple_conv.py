"""The four-variable model as the stand-alone scripts of the reference write it: the no-flux boundary spelled out as
a SYMMETRIC re-pad of the interior, the Laplacian as a 3x3 single-channel convolution with zero ('SAME') padding, one
sub-step per tick.  Exercises `tf.pad`, `tf.expand_dims` and `tf.nn.depthwise_conv2d` of the tracer; the kernel
generated from it must equal the hand-written FIBHIP_ZEROPAD kernel bit for bit under the rounding-faithful policy
(tests/test_gpu_traced.py::test_generated_conv_variant_equals_handwritten)."""
import numpy as np
import tensorflow as tf

from .four_variable import FourVariable


def no_flux(X):
    """border rows and columns take the values next to them"""
    return tf.pad(X[1:-1, 1:-1], tf.constant([[1, 1], [1, 1]]), 'SYMMETRIC', name='no_flux')


def conv_laplacian(X):
    weights = np.array([[0.5, 1.0, 0.5], [1.0, -6.0, 1.0], [0.5, 1.0, 0.5]])
    kernel = tf.constant(weights.reshape(3, 3, 1, 1), dtype=tf.float32)
    sheet = tf.expand_dims(tf.expand_dims(X, 0), -1)                  # [1, H, W, 1]
    return tf.nn.depthwise_conv2d(sheet, kernel, [1, 1, 1, 1], padding='SAME')[0, :, :, 0]


class FourVariableConv(FourVariable):
    def solve(self, state):
        U, V, W, S = state
        U0 = no_flux(U)
        dU, dV, dW, dS = self.rates(U, V, W, S)
        U1 = U0 + self.dt * dU + self.diff * self.dt * conv_laplacian(U0)
        return U1, V + self.dt * dV, W + self.dt * dW, S + self.dt * dS

    def define(self, s1=True):
        super(FourVariable, self).define()
        shape = [self.height, self.width]
        u = np.zeros(shape, dtype=np.float32)
        if s1:
            u[:, 1] = 1.0
        vars_ = [tf.Variable(u, name='U'), tf.Variable(np.ones(shape, dtype=np.float32), name='V'),
                 tf.Variable(np.ones(shape, dtype=np.float32), name='W'),
                 tf.Variable(np.zeros(shape, dtype=np.float32), name='S')]
        new = self.solve(tuple(vars_))
        self.dt_per_step = 1
        self._ode_op = tf.group(*[tf.assign(v, n) for v, n in zip(vars_, new)])
        self._U = vars_[0]
