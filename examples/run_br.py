#!/usr/bin/env python3
"""The reference's `br.py __main__` (br.py:347-382) against fib_tf_amd."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.br import BeelerReuter

if __name__ == '__main__':
    config = {'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 0.809, 'duration': 1000,
              'skip': False, 'cheby': True, 'timeline': False, 'timeline_name': 'timeline_br.json',
              'save_graph': False}
    model = BeelerReuter(config)
    model.add_hole_to_phase_field(150, 200, 40)
    model.define()
    model.add_pace_op('s2', 'luq', 10.0)
    s2 = model.millisecond_to_step(300)
    ds = model.millisecond_to_step(10)
    n = int(model.duration / 10.0)
    cube = np.zeros([n, model.height, model.width], dtype=np.float32)
    for i in model.run(None):
        if i == s2:
            model.fire_op('s2')
        if i % ds == 0:
            cube[i // ds, :, :] = model.image() * model.phase
    np.save('cube', cube)
