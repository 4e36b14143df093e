#!/usr/bin/env python3
"""The reference's `court.py __main__` protocol (court.py:585-636), shortened: fast tick every iteration,
'slow' + 'trend' every 10th, S2 at 350 ms, keep_state -> define(state=...) resume."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.court import Courtemanche, cl_observer

if __name__ == '__main__':
    config = {'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 0.809,
              'duration': float(sys.argv[1]) if len(sys.argv) > 1 else 500, 'skip': False, 'cheby': True,
              'timeline': False, 'timeline_name': 'timeline_court.json', 'save_graph': False}
    m1 = Courtemanche(config)
    m1.add_hole_to_phase_field(256, 256, 30)
    m1.add_hole_to_phase_field(256, 256, 250, neg=True)
    m1.define()
    m1.add_pace_op('s2', 'luq', 10.0)
    m1.cl_observer = cl_observer
    s2 = m1.millisecond_to_step(350)
    data = []
    for i in m1.run(None, keep_state=True, block=False):
        if i % 10 == 0:
            m1.fire_op('slow')
            m1.fire_op('trend')
            data.append(m1._Trend.eval())
        if i == s2:
            m1.fire_op('s2')

    m2 = Courtemanche(dict(config, duration=50))
    m2.add_hole_to_phase_field(256, 256, 100)
    m2.add_hole_to_phase_field(256, 256, 250, neg=True)
    m2.define(state=m1.state)
    for i in m2.run(None):
        if i % 10 == 0:
            m2.fire_op('slow')
            m2.fire_op('trend')
            data.append(m2._Trend.eval())
    np.savetxt('vol_na_2.dat', np.asarray(data))
