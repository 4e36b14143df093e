"""not-gpu: the model-file tracer (fib_tf_amd.tfgraph + fib_tf_amd.traced) on the CPU.

(1) The reference's UNCHANGED fenton.py / br.py / court.py are imported with tfgraph installed as `tensorflow`
    (only in this container: /root/reference does not travel), traced, and the recorded graphs interpreted
    op by op (oracle/graph_eval.py) against the committed golden trajectories — this pins both the tracer and
    the interpreter that the GPU tests use as the oracle for traced models.
(2) Structure of the generated HIP source and the tracer's error behaviour, on the model files of tests/models/.
"""
import importlib
import os
import sys

import numpy as np
import pytest

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture
def ref_modules():
    if not os.path.isdir(REF):
        pytest.skip('the reference tree is only mounted in the build container')
    import fib_tf_amd.tfgraph as tfg
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'ionic', 'screen', 'fenton', 'br', 'court')}
    old_dwb = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                     # the reference tree is read-only
    tfg.install()
    sys.path.append(REF)
    for k in ('fenton', 'br', 'court'):
        sys.modules.pop(k, None)
    try:
        yield {k: importlib.import_module(k) for k in ('fenton', 'br', 'court')}
    finally:
        sys.path.remove(REF)
        sys.dont_write_bytecode = old_dwb
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def cfg(H, W, diff, **kw):
    c = {'width': W, 'height': H, 'dt': 0.1, 'dt_per_plot': 10, 'diff': diff, 'duration': 100, 'timeline': False,
         'timeline_name': 'timeline.json', 'save_graph': False, 'skip': False, 'cheby': True}
    c.update(kw)
    return c


def state_of(f, names):
    return np.stack([f['init_' + n] for n in names])


def check(state, f, names, t, tol, scales=None):
    for i, n in enumerate(names):
        want = f['%s_t%d' % (n, t)]
        s = (scales or {}).get(n, max(1.0, float(np.abs(want).max())))
        err = float(np.abs(state[i] - want).max())
        assert err <= tol * s, '%s t%d: max|d| %.3e > %.1e*%g' % (n, t, err, tol, s)


def test_reference_fenton_file_traced(ref_modules, golden):
    from oracle.graph_eval import Interpreter
    f = golden('fenton_traj64')
    H, W = f['phase'].shape
    m = ref_modules['fenton'].Fenton4v(cfg(H, W, float(f['diff'])))
    m.add_hole_to_phase_field(*[int(x) for x in f['hole']])
    assert np.array_equal(m.phase, f['phase'])
    m.define()
    c = m._analyze()
    assert c['spt'] == 10 and m.dt_per_step == 10 and m.VAR_NAMES == ('U', 'V', 'W', 'S')
    # ten chained solve() calls are ONE sub-step function repeated: fused 10 deep
    assert len({lv.signature() for lv in c['programs'][0][1].levels}) == 1
    assert '#define FIB_CUSTOM_K 10' in c['source'] and ('P::tanhv' in c['source'] or 'P::one_plus_tanh' in c['source']) and 'vsel(' in c['source']
    it = Interpreter(c, m.phase)
    st = state_of(f, m.VAR_NAMES)
    assert np.array_equal(st, np.stack([v.init for v in c['slots']]))      # define()'s own S1 initial state
    t0 = 0
    for t in [1, 2, 10, 20]:
        st = it.tick(st, t - t0)
        t0 = t
        check(st, f, m.VAR_NAMES, t, 1e-6)


@pytest.mark.parametrize('name,cheby,skip', [('br_traj64_cheby', True, False), ('br_traj64_direct', False, False),
                                             ('br_traj64_skip', False, True)])
def test_reference_br_file_traced(ref_modules, golden, name, cheby, skip):
    from oracle.graph_eval import Interpreter
    f = golden(name)
    H, W = f['phase'].shape
    m = ref_modules['br'].BeelerReuter(cfg(H, W, float(f['diff']), cheby=cheby, skip=skip))
    m.add_hole_to_phase_field(*[int(x) for x in f['hole']])
    m.define()
    c = m._analyze()
    names = ('V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI')
    assert m.VAR_NAMES == names and c['spt'] == 5
    kinds = [lv.signature() for lv in c['programs'][0][1].levels]
    assert len(set(kinds)) == (2 if skip else 1)           # skip: solve(.,5) once, then solve(.,0) four times
    if skip:
        assert kinds[0] != kinds[1] and len(set(kinds[1:])) == 1 and 'if (sub == 0)' in c['source']
    it = Interpreter(c, m.phase)
    st = state_of(f, names)
    t0 = 0
    for t in [1, 4, 20]:
        st = it.tick(st, t - t0)
        t0 = t
        check(st, f, names, t, 2e-6, {'V': 120.0, 'C': 1e-4})


def test_reference_court_file_traced(ref_modules, golden):
    from oracle.graph_eval import Interpreter
    f = golden('court_traj64')
    H, W = f['phase'].shape
    m = ref_modules['court'].Courtemanche(cfg(H, W, float(f['diff'])))
    m.phase = f['phase']
    m.define()
    c = m._analyze()
    names = [str(n) for n in f['names']]
    assert len(c['programs']) == 2 and c['programs'][1][0] == 'slow'
    fast, slow = c['programs'][0][1], c['programs'][1][1]
    assert {m.VAR_NAMES[c['remap'][p]] for p in fast.mask} == {'V', '_Na_i_', '_m_', '_h_'}
    assert len(slow.mask) == 17 and not slow.uses_lap and fast.uses_lap
    assert m.VAR_NAMES[0] == 'V' and set(m.VAR_NAMES) == set(names)
    it = Interpreter(c, m.phase)
    st = np.stack([f['init_' + n] for n in m.VAR_NAMES])
    t0 = 0
    for t in [1, 2, 10, 11]:
        for i in range(t0, t):                             # the reference driver: court.py:612-617
            st = it.run_mode(st, 0)
            if i % 10 == 0:
                st = it.run_mode(st, 1)
        t0 = t
        check(st, f, m.VAR_NAMES, t, 2e-6, {'V': 150.0, '_Ca_i_': 1e-3})
    # the Trend probe (court.py:107-111) is a host-side op with two element assigns
    mode, host = c['modes'][id(m._ops['trend'])]
    assert mode is None and len(host) == 2
