#!/usr/bin/env python3
"""The reference's `fenton_simple.py __main__` (fenton_simple.py:223-239; `fenton_jit.py` is the same with another
class name): 512x512, 10 000 steps, S2 after step 2100, a frame every 10 steps into a (headless) Screen."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.fenton_simple import Fenton4vSimple
from fib_tf_amd.screen import Screen

if __name__ == '__main__':
    config = {
        'width': 512,
        'height': 512,
        'dt': 0.1,
        'dt_per_plot': 10,
        'diff': 1.5,
        'samples': int(sys.argv[1]) if len(sys.argv) > 1 else 10000,
        's2_time': 210
    }
    model = Fenton4vSimple(config)
    model.define()
    # note: pass None instead of a Screen to run without frames (then 10 steps are fused per launch)
    im = Screen(model.height, model.width, 'Simple Fenton 4v Model')
    model.run(im)
    im.save('fenton_simple.png')
    print('%d frames; %.0f Mcell-steps/s including them'
          % (im.count, model.height * model.width * model.samples / model.elapsed / 1e6))
