#!/usr/bin/env python3
"""One simulated second of spiral-wave re-entry in the four-variable atrial model on a 512 x 512 sheet with a
circular obstacle: a planar wave is started from the left edge (S1, part of the initial state), a second stimulus
in the upper-left quadrant 210 ms later breaks it, and the broken end curls around the obstacle.  Every 10 ms
the potential is read back (masked by the phase field) into a [frames, H, W] array saved as cube.npy, the format
`python -m fib_tf_amd.playcube` replays.  Drives fib_tf_amd through the IonicModel / define() / run() interface.

    python examples/run_fenton.py [--size N] [--ms T] [--frames DIR]
        --size N      sheet edge in cells (default 512; obstacle and quadrant scale with it)
        --ms T        simulated milliseconds (default 1000)
        --frames DIR  also write PNG frames through the headless Screen
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.fenton import Fenton4v
from fib_tf_amd.screen import Screen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ms', type=float, default=1000.0)
    ap.add_argument('--frames', default=None)
    args = ap.parse_args()
    n = args.size
    sheet = Fenton4v({'width': n, 'height': n, 'dt': 0.1, 'diff': 1.5, 'duration': args.ms, 'dt_per_plot': 10,
                      'timeline': False, 'timeline_name': 'timeline_4v.json', 'save_graph': False})
    sheet.add_hole_to_phase_field(n // 2, n // 2, 30 * n / 512.0)
    sheet.define()
    sheet.add_pace_op('s2', 'luq', 1.0)

    screen = None
    if args.frames:
        os.makedirs(args.frames, exist_ok=True)
        screen = Screen(n, n, 'four-variable model', png_pattern=os.path.join(args.frames, 'frame_%05d.png'))

    second_stimulus = sheet.millisecond_to_step(210)
    stride = sheet.millisecond_to_step(10)              # ticks between two recorded frames
    frames = np.zeros([int(args.ms / 10.0), n, n], dtype=np.float32)
    for tick in sheet.run(screen):
        if tick == second_stimulus:
            sheet.fire_op('s2')
        if tick % stride == 0 and tick // stride < len(frames):
            frames[tick // stride] = sheet.image() * sheet.phase
    np.save('cube', frames)
    cell_steps = n * n * sheet.samples * sheet.dt_per_step
    print('%.0f Mcell-steps/s including the %d read-backs' % (cell_steps / sheet.elapsed / 1e6, len(frames)))


if __name__ == '__main__':
    main()
