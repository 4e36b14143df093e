"""TEST INFRASTRUCTURE.  Traces the reference's UNCHANGED model files (fenton.py, br.py, court.py under
/root/reference) with fib_tf_amd.tfgraph / fib_tf_amd.traced and compiles the generated HIP into
oracle/_ref/traced/<case>.so (+ <case>.json: variable order, sub-steps per tick, assign-group masks).

Like oracle/_ref/generate_table this is a build of the reference's own sources where they lie: outputs are
git-ignored and travel to the GPU box as binaries; nothing of the reference is copied into the repository (the
generated header is written to a temporary directory and removed).  tests/test_gpu_traced_reference.py runs
these libraries against the golden fixtures and against the hand-written kernels."""
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.path.join(HERE, '_ref', 'traced')

# case -> (module, class, config overrides); dt = 0.1 everywhere (the fixtures' value)
CASES = {
    'fenton_d1.5': ('fenton', 'Fenton4v', {'diff': 1.5}),
    'fenton_d1.1': ('fenton', 'Fenton4v', {'diff': 1.1}),
    'br_cheby_d0.809': ('br', 'BeelerReuter', {'diff': 0.809, 'cheby': True, 'skip': False}),
    'br_direct_d0.809': ('br', 'BeelerReuter', {'diff': 0.809, 'cheby': False, 'skip': False}),
    'br_skip_d0.809': ('br', 'BeelerReuter', {'diff': 0.809, 'cheby': False, 'skip': True}),
    'court_d0.809': ('court', 'Courtemanche', {'diff': 0.809}),
}


def build(verbose=False):
    if not os.path.isdir(REF):
        return []
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import importlib
    import fib_tf_amd.tfgraph as tfg
    from fib_tf_amd import _lib, traced
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'ionic', 'screen', 'fenton', 'br', 'court')}
    dwb = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    tfg.install()
    sys.path.append(REF)
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix='fib_traced_')
    built = []
    try:
        for k in ('fenton', 'br', 'court'):
            sys.modules.pop(k, None)
        for case, (mod, cls, over) in CASES.items():
            so = os.path.join(OUT, case + '.so')
            meta = os.path.join(OUT, case + '.json')
            if os.path.exists(so) and os.path.exists(meta) and \
                    all(os.path.getmtime(so) >= os.path.getmtime(d) for d in _lib.DEPS + [__file__, traced.__file__, tfg.__file__]):
                built.append(so)
                continue
            cfg = {'width': 32, 'height': 32, 'dt': 0.1, 'dt_per_plot': 10, 'diff': 1.0, 'duration': 10,
                   'timeline': False, 'timeline_name': 'timeline.json', 'save_graph': False, 'skip': False,
                   'cheby': True}
            cfg.update(over)
            m = getattr(importlib.import_module(mod), cls)(cfg)
            m.define()
            c = m._analyze()
            inc = os.path.join(tmp, case + '.inc')
            with open(inc, 'w') as f:
                f.write(c['source'])
            _lib.build_custom(inc, so, verbose=verbose)
            info = {'names': list(m.VAR_NAMES), 'spt': c['spt'], 'dt': cfg['dt'], 'diff': cfg['diff'],
                    'modes': [{'name': n, 'mask': sorted(c['remap'][p] for p in prog.mask)}
                              for n, prog in c['programs']]}
            with open(meta, 'w') as f:
                json.dump(info, f)
            built.append(so)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        sys.path.remove(REF)
        sys.dont_write_bytecode = dwb
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return built


if __name__ == '__main__':
    for p in build(verbose='-v' in sys.argv):
        print('built', os.path.relpath(p, ROOT))
