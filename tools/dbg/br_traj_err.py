"""Diagnostic: the error of the Beeler-Reuter fast policy along the golden 64x64 trajectories (tests/golden/br_traj64_*.npz), per array
and snapshot tick, for the build of the library named by FIBHIP_BR_LIBRARY (tools/r04_p.sh compares the FIB_BR_FEWER levels)."""
import os, sys
import numpy as np
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from fib_tf_amd.br import BeelerReuter

def cfg(h, w, diff, **kw):
    c = {'width': w, 'height': h, 'dt': 0.1, 'dt_per_plot': 10, 'diff': diff, 'duration': 1000, 'timeline': False,
         'timeline_name': 'unused.json', 'save_graph': False, 'fast_math': True}
    c.update(kw)
    return c

for name in sys.argv[1:] or ['br_traj64_cheby', 'br_traj64_cheby_skip']:
    f = np.load(os.path.join(root, 'tests', 'golden', name + '.npz'))
    m = BeelerReuter(cfg(64, 64, float(f['diff']), cheby=bool(f['cheby']), skip=bool(f['skip'])))
    m.add_hole_to_phase_field(*[float(x) for x in f['hole']])
    m.define()
    m.add_pace_op('s2', 'luq', 10.0)
    t0 = 0
    for t in [int(x) for x in f['snap_ticks']]:
        m.duration = (t - t0) * m.dt_per_step * m.dt + 1e-9
        for i in m.run():
            if name.endswith('cheby_skip') and i + t0 == 10:
                m.fire_op('s2')
        t0 = t
        row = []
        for k in m.VAR_NAMES:
            want = f['%s_t%d' % (k, t)]
            scale = {'V': 120.0, 'C': max(float(want.max() - want.min()), float(np.abs(want).max()))}.get(k, 1.0)
            row.append('%s %.1e' % (k, float(np.abs(m._State[k].eval().astype(np.float64) - want).max()) / scale))
        print(name, 't%d' % t, '  '.join(row))
