import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from fib_tf_amd.br import BeelerReuter
f = np.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests/golden/br_step.npz'))
CFG = {'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000, 'timeline': False, 'timeline_name': 'unused.json', 'save_graph': False, 'skip': False}
for cheby in (False, True):
  for spec in (True, False):
    m = BeelerReuter(dict(CFG, height=37, width=53, diff=0.809, fast_math=True, cheby=cheby, specialise=spec))
    m.phase = f['phase']
    names = m.VAR_NAMES
    for n in (1, 5):
        out = m.solve(tuple(f[k] for k in names), n)
        tag = 'cheby' if cheby else 'direct'
        worst = 0
        for k, o in zip(names, out):
            want = f['%s1_%s_n%d' % (k, tag, n)]
            scale = {'V': 120.0, 'C': 1e-5}.get(k, 1.0)
            e = float(np.abs(np.asarray(o, np.float64) - want).max()) / scale
            worst = max(worst, e)
            print(tag, 'spec' if spec else 'stock', 'n', n, k, '%.2e' % e)
        print('   worst %.2e' % worst)
