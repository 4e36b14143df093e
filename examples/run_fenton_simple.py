#!/usr/bin/env python3
"""The stand-alone four-variable script variant (zero-padded 3x3 convolution Laplacian, built-in second stimulus):
N single steps on a 512 x 512 sheet, every tenth one painted into a headless Screen whose last frame is saved as
fenton_simple.png.

    python examples/run_fenton_simple.py [steps] [--no-frames]
        --no-frames   run without a Screen (ten steps are then fused per launch)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.fenton_simple import Fenton4vSimple
from fib_tf_amd.screen import Screen


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    steps = int(argv[0]) if argv else 10000
    sheet = Fenton4vSimple({'width': 512, 'height': 512, 'dt': 0.1, 'diff': 1.5, 'dt_per_plot': 10,
                            'samples': steps, 's2_time': 210})
    sheet.define()
    screen = None if '--no-frames' in sys.argv else Screen(sheet.height, sheet.width, 'four-variable model, simple variant')
    sheet.run(screen)
    rate = sheet.height * sheet.width * sheet.samples / sheet.elapsed / 1e6
    if screen is None:
        print('%.0f Mcell-steps/s' % rate)
    else:
        screen.save('fenton_simple.png')
        print('%d frames; %.0f Mcell-steps/s including them' % (screen.count, rate))


if __name__ == '__main__':
    main()
