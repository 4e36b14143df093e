#!/usr/bin/env python3
"""The reference's `fenton.py __main__` (fenton.py:155-187) against fib_tf_amd: identical driver code,
only the import differs.  Writes cube.npy (100 frames of image()*phase) like the reference.

    python examples/run_fenton.py [--frames DIR]     # --frames: also write PNG frames through the headless Screen
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.fenton import Fenton4v          # reference: from fenton import Fenton4v
from fib_tf_amd.screen import Screen            # reference: from screen import Screen

if __name__ == '__main__':
    config = {
        'width': 512,           # screen width in pixels
        'height': 512,          # screen height in pixels
        'dt': 0.1,              # integration time step in ms
        'dt_per_plot': 10,      # screen refresh interval in dt unit
        'diff': 1.5,            # diffusion coefficient
        'duration': 1000,       # simulation duration in ms
        'timeline': False,      # flag to save a timeline (profiler)
        'timeline_name': 'timeline_4v.json',
        'save_graph': True      # accepted, ignored (there is no TF graph)
    }
    model = Fenton4v(config)
    model.add_hole_to_phase_field(256, 256, 30)
    model.define()
    model.add_pace_op('s2', 'luq', 1.0)
    im = None
    if '--frames' in sys.argv:
        d = sys.argv[sys.argv.index('--frames') + 1]
        os.makedirs(d, exist_ok=True)
        im = Screen(model.height, model.width, 'Fenton 4v Model', png_pattern=os.path.join(d, 'frame_%05d.png'))

    s2 = model.millisecond_to_step(210)     # 210 ms
    ds = model.millisecond_to_step(10)
    n = int(model.duration / 10.0)
    cube = np.zeros([n, model.height, model.width], dtype=np.float32)

    for i in model.run(im):
        if i == s2:
            model.fire_op('s2')
        if i % ds == 0:
            cube[i // ds, :, :] = model.image() * model.phase

    np.save('cube', cube)
    print('%.0f Mcell-steps/s (including the %d image() read-backs)' % (
        model.height * model.width * model.samples * model.dt_per_step / model.elapsed / 1e6, n))
