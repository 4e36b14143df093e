#!/usr/bin/env python3
"""Spiral-wave re-entry in the eight-variable ventricular model (Beeler-Reuter) around a circular obstacle.

A plane wave leaves the left edge at t = 0 (it is part of the initial state); while its tail crosses the upper-left
quadrant a second stimulus is applied there, which turns the wave back on itself.  The potential, masked by the phase
field, is sampled every `--every` milliseconds into cube.npy (`python -m fib_tf_amd.playcube cube.npy` replays it) and
the arrival times at two probe electrodes are printed, from which the conduction velocity follows.

    python examples/run_br.py [--size N] [--ms T] [--s2 T2] [--every E] [--direct] [--skip] [--out FILE]
        --direct   evaluate the gates with exponentials instead of the Chebyshev fits
        --skip     advance the slow gates once per tick (multirate schedule)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.br import BeelerReuter


def build_sheet(args):
    n, scale = args.size, args.size / 512.0
    sheet = BeelerReuter({'width': n, 'height': n, 'dt': 0.1, 'diff': 0.809, 'duration': args.ms, 'dt_per_plot': 10,
                          'cheby': not args.direct, 'skip': args.skip, 'timeline': False,
                          'timeline_name': 'timeline_br.json', 'save_graph': False})
    sheet.add_hole_to_phase_field(150 * scale, 200 * scale, 40 * scale)
    sheet.define()
    sheet.add_pace_op('second', 'luq', 10.0)
    return sheet


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ms', type=float, default=1000.0)
    ap.add_argument('--s2', type=float, default=300.0, help='time of the second stimulus, ms')
    ap.add_argument('--every', type=float, default=10.0, help='sampling period of the recorded frames, ms')
    ap.add_argument('--direct', action='store_true')
    ap.add_argument('--skip', action='store_true')
    ap.add_argument('--out', default='cube.npy')
    args = ap.parse_args()

    sheet = build_sheet(args)
    n = args.size
    tick_s2 = sheet.millisecond_to_step(args.s2)
    period = max(1, sheet.millisecond_to_step(args.every))
    frames = []
    probes = [(n // 2, n // 4), (n // 2, n // 2)]            # two electrodes on the middle row
    arrival = [None, None]
    for tick in sheet.run():
        if tick == tick_s2:
            sheet.fire_op('second')
        if tick % period == 0:
            frame = sheet.image()
            frames.append(frame * sheet.phase)
            for k, (r, c) in enumerate(probes):
                if arrival[k] is None and frame[r, c] > 0.5:
                    arrival[k] = tick * sheet.dt * sheet.dt_per_step
    np.save(args.out, np.asarray(frames, dtype=np.float32))
    print('%d frames of %dx%d written to %s' % (len(frames), n, n, args.out))
    if None not in arrival and arrival[1] > arrival[0]:
        print('wave front: column %d at %.0f ms, column %d at %.0f ms -> %.2f cells/ms' % (
            probes[0][1], arrival[0], probes[1][1], arrival[1], (probes[1][1] - probes[0][1]) / (arrival[1] - arrival[0])))
    print('%.0f Mcell-steps/s including the read-backs' % (n * n * sheet.samples * sheet.dt_per_step / sheet.elapsed / 1e6))


if __name__ == '__main__':
    main()
