#!/usr/bin/env python3
"""Atrial re-entry in the 21-variable Courtemanche model, in two stages, with the state handed from one to the next.

Stage 1: a ring of tissue (everything outside a large disc and inside a small one is removed from the phase field); a
wave is started at the left, a second stimulus breaks it.  Every tick assigns the four fast variables (V, Na_i and the
two sodium gates); the other seventeen follow every tenth tick through the model's 'slow' operation, which is when the
probe trace (potential and sodium concentration at the sheet's centre row) is sampled too.
Stage 2: the final state of stage 1 continues on a sheet with a LARGER central obstacle (define(state=...)), the way a
protocol changes the substrate under a running arrhythmia.

    python examples/run_court.py [--size N] [--ms T] [--ms2 T] [--healthy] [--out FILE]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.court import Courtemanche, cl_observer

SLOW_EVERY = 10        # ticks between two 'slow' updates (the model integrates its slow set with 10 dt)


def ring(sheet, inner, outer):
    c = sheet.width / 2.0
    sheet.add_hole_to_phase_field(c, c, inner)
    sheet.add_hole_to_phase_field(c, c, outer, neg=True)


def advance(sheet, trace, stimulus_tick=None, keep=False):
    for tick in sheet.run(None, keep_state=keep, block=False):
        if tick % SLOW_EVERY == 0:
            sheet.fire_op('slow')
            sheet.fire_op('trend')
            trace.append(sheet._Trend.eval())
        if tick == stimulus_tick:
            sheet.fire_op('break')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ms', type=float, default=500.0, help='duration of stage 1')
    ap.add_argument('--ms2', type=float, default=50.0, help='duration of stage 2')
    ap.add_argument('--healthy', action='store_true',
                    help='normal conductances (the model class defaults to the chronically remodelled parameter set)')
    ap.add_argument('--out', default='court_trace.txt')
    args = ap.parse_args()
    n, scale = args.size, args.size / 512.0
    common = {'width': n, 'height': n, 'dt': 0.1, 'diff': 0.809, 'dt_per_plot': 10, 'timeline': False,
              'timeline_name': 'timeline_court.json', 'save_graph': False}

    first = Courtemanche(dict(common, duration=args.ms))
    first.chronic = not args.healthy
    ring(first, 30 * scale, 250 * scale)
    first.define()
    first.add_pace_op('break', 'luq', 10.0)
    first.cl_observer = cl_observer
    trace = []
    advance(first, trace, stimulus_tick=first.millisecond_to_step(350), keep=True)

    second = Courtemanche(dict(common, duration=args.ms2))
    second.chronic = not args.healthy
    ring(second, 100 * scale, 250 * scale)
    second.define(state=first.state)
    advance(second, trace)

    np.savetxt(args.out, np.asarray(trace))
    print('%d probe samples (every %d ticks) written to %s; stage 1 ran %.0f Mcell-steps/s' % (
        len(trace), SLOW_EVERY, args.out, n * n * first.samples / first.elapsed / 1e6))


if __name__ == '__main__':
    main()
