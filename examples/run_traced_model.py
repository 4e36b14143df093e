"""A model of your own in the reference's style, no device code: the Karma two-variable model (Chaos 4:461, 1994)
written against fib_tf_amd.tfgraph, traced by fib_tf_amd.traced and run as one fused HIP launch per tick.

    python examples/run_traced_model.py            # prints the generated HIP source size, runs 300 ms, writes karma.png
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fib_tf_amd.tfgraph as tf                    # noqa: E402   (where a reference model file says `import tensorflow as tf`)
from fib_tf_amd.screen import Screen               # noqa: E402
from fib_tf_amd.traced import IonicModel           # noqa: E402   (... `from ionic import IonicModel`)


class Karma(IonicModel):
    def __init__(self, props):
        IonicModel.__init__(self, props)
        self.min_v = 0.0
        self.max_v = 4.0
        self.depol = 0.0

    def solve(self, state):
        E, n = state
        eps, gamma, beta, E_star, E_n, M = 0.01, 1.1, 1.389, 1.5415, 1.0, 4
        E0 = self.enforce_boundary(E)
        theta = (1 + tf.sign(E - E_n)) * 0.5                      # Heaviside, the way fenton.py writes it
        R = 1.0 / (1.0 - np.exp(-beta))
        f = -E + (E_star - tf.pow(n, M)) * (1 - tf.tanh(E - 3.0)) * tf.square(E) * 0.5
        dn = eps * (R * theta - n)
        E1 = E0 + self.dt * (f / 2.5) + self.diff * self.dt * self.laplace(E0)
        n1 = n + self.dt * dn
        return E1, n1

    def define(self, s1=True):
        super().define()
        e0 = np.zeros([self.height, self.width], dtype=np.float32)
        n0 = np.zeros([self.height, self.width], dtype=np.float32)
        if s1:
            e0[:, :4] = 3.0
        E = tf.Variable(e0, name='E')
        n = tf.Variable(n0, name='n')
        states = [(E, n)]
        for i in range(10):
            states.append(self.solve(states[-1]))
        E1, n1 = states[-1]
        self.dt_per_step = 10
        self._ode_op = tf.group(E.assign(E1), n.assign(n1))
        self._E = E

    def pot(self):
        return self._E

    def image(self):
        return (self._E.eval() - self.min_v) / (self.max_v - self.min_v)


if __name__ == '__main__':
    config = {'width': 512, 'height': 512, 'dt': 0.05, 'dt_per_plot': 10, 'diff': 1.0, 'duration': 300}
    model = Karma(config)
    model.add_hole_to_phase_field(256, 256, 30)
    model.define()
    print('generated HIP source: %d lines' % len(model.generated_source().splitlines()))
    model.add_pace_op('s2', 'luq', 3.0)
    s2 = model.millisecond_to_step(170)
    im = Screen(model.height, model.width, 'Karma model (traced)')
    for i in model.run(im, block=False):
        if i == s2:
            model.fire_op('s2')
    fused, launches = model._stepper.launch_plan()
    cells = model.height * model.width * model.samples * model.dt_per_step
    print('%d sub-steps fused per launch, %d launch(es) per tick; %.0f Mcell-steps/s incl. %d frames'
          % (fused, launches, cells / model.elapsed / 1e6, im.count))
    im.save('karma.png')
