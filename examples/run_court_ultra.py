#!/usr/bin/env python3
"""The reference's `court_ultra.py __main__` protocol (court_ultra.py:530-559), shortened: the single-rate 21- (or,
with 'ultra_slow', 22-) variable model on an annular domain, S1-S2, the phase-weighted cycle-length observer, state
checkpoint `state_small.npy`, then a second stage resumed from it around a larger obstacle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd.court_ultra import run_large, run_small
from fib_tf_amd.screen import Screen

if __name__ == '__main__':
    config = {
        'width': 512, 'height': 512, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 1.5,
        'duration': float(sys.argv[1]) if len(sys.argv) > 1 else 600,     # the reference runs 10 000 ms
        'skip': False, 'cheby': True, 'timeline': False, 'timeline_name': 'timeline_court.json', 'save_graph': False,
        'ultra_slow': len(sys.argv) > 2 and sys.argv[2] == 'ultra_slow',
    }
    im = Screen(config['height'], config['width'], 'Courtemanche Model')
    cyclelengths = []
    run_small(config, im, cyclelengths, radius=10)
    i0 = int(config['duration'] / config['dt'])
    run_large(dict(config, duration=100), im, cyclelengths, 100, i0)
    im.save('100.png')
    print('%d frames, %d cycle-length records' % (im.count, len(cyclelengths)))
