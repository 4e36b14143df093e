#!/usr/bin/env python3
"""GPU box: max |d| of both arithmetic policies against the golden trajectories (for choosing / stating tolerances)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fib_tf_amd.fenton import Fenton4v
from fib_tf_amd.br import BeelerReuter
from fib_tf_amd.court import Courtemanche
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
base = {'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000}
def adv(m, n, hook=None, t0=0):
    m.duration = n * m.dt_per_step * m.dt + 1e-9
    for i in m.run():
        if hook: hook(i + t0)
for fast in (False, True):
    f = np.load(G + '/fenton_traj64.npz')
    m = Fenton4v(dict(base, height=64, width=64, diff=1.5, fast_math=fast)); m.add_hole_to_phase_field(32, 32, 6); m.define()
    t0 = 0
    for t in f['snap_ticks']:
        adv(m, int(t) - t0); t0 = int(t)
        print('fenton fast=%d tick %3d:' % (fast, t), ' '.join('%s %.2e' % (k, np.abs(m._State[k].eval().astype(np.float64) - f['%s_t%d' % (k, t)]).max()) for k in 'UVWS'))
    for name in ('br_traj64_direct', 'br_traj64_cheby'):
        f = np.load(G + '/%s.npz' % name)
        m = BeelerReuter(dict(base, height=64, width=64, diff=0.809, cheby=bool(f['cheby']), skip=False, fast_math=fast)); m.add_hole_to_phase_field(20, 30, 6); m.define()
        t0 = 0
        for t in f['snap_ticks']:
            adv(m, int(t) - t0); t0 = int(t)
            print('%s fast=%d tick %3d:' % (name, fast, t), ' '.join('%s %.2e' % (k, np.abs(m._State[k].eval().astype(np.float64) - f['%s_t%d' % (k, t)]).max()) for k in ('V', 'M', 'H', 'D', 'XI')))
    f = np.load(G + '/court_traj64.npz')
    m = Courtemanche(dict(base, height=64, width=64, diff=0.809, fast_math=fast)); m.phase = f['phase']; m.define()
    t0 = 0
    def hook(i):
        if i % 10 == 0: m.fire_op('slow')
    for t in f['snap_ticks']:
        adv(m, int(t) - t0, hook, t0); t0 = int(t)
        print('court fast=%d tick %3d:' % (fast, t), ' '.join('%s %.2e' % (k, np.abs(m._State[k].eval().astype(np.float64) - f['%s_t%d' % (k, t)]).max()) for k in ('V', '_m_', '_Na_i_', '_Ca_i_', '_Ca_rel_', '_u_')))
