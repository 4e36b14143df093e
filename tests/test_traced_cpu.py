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


def test_reference_files_generate_compilable_source(ref_modules, tmp_path):
    """compile-only: the HIP source generated from the reference's unchanged fenton.py and br.py builds for gfx950
    (hipcc cross-compiles here).  The binaries stay in the test's temporary directory: nothing derived from the
    reference leaves this container."""
    from fib_tf_amd import _lib
    for mod, cls, over in (('fenton', 'Fenton4v', {}), ('br', 'BeelerReuter', {'cheby': True})):
        m = getattr(ref_modules[mod], cls)(cfg(32, 32, 1.0, **over))
        m.define()
        inc = tmp_path / (mod + '.inc')
        inc.write_text(m._analyze()['source'])
        so = tmp_path / (mod + '.so')
        _lib.build_custom(str(inc), str(so))
        assert so.stat().st_size > 10000


# ---------------------------------------------------------------------------------------------------------
# (2) no reference tree needed: tfgraph semantics, tracer structure and error behaviour on tests/models/
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture
def clean_modules():
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'ionic', 'screen')}
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_tfgraph_scalar_semantics():
    """a Python / NumPy scalar becomes float32 where it meets a tensor (TF's conversion rule); Python-side products
    stay double; NumPy scalars on the left defer to the tensor's reflected operator"""
    import fib_tf_amd.tfgraph as tf
    x = tf.Variable(np.zeros((4, 4)))
    e = 0.1 * x
    assert e.op == 'mul' and e.args[0] == float(np.float32(0.1)) and e.args[1] is x
    e = np.float64(1.5) * 0.1 - x                     # the product is formed in double first (fenton.py:103)
    assert e.op == 'sub' and e.args[0] == float(np.float32(1.5 * 0.1))
    assert (np.float64(2.0) / x).op == 'div' and (x ** 2).op == 'pow' and (-x).op == 'neg'
    assert tf.square(0.0337) == float(np.square(np.float32(0.0337)))            # court.py:310
    assert (x > 0.5).is_mask and tf.where(x > 0.5, x, 0.0).op == 'where'
    with pytest.raises(TypeError):
        x + np.zeros((4, 4))                          # array constants belong in a tf.Variable
    with pytest.raises(TypeError):
        tf.where(x, x, x)                             # condition must be a comparison
    g = tf.group(x.assign(x + 1), tf.group(tf.assign(x, x * 2)), None)
    assert len(g) == 2
    assert np.array_equal(x.eval(), np.zeros((4, 4), np.float32))               # before compile: the initial value


@pytest.mark.parametrize('name,fixture,ticks,tol,scales', [
    ('fv', 'fenton_traj64', [1, 2, 10, 20], 1e-6, None),
    ('ev', 'br_traj64_direct', [1, 4, 20], 2e-6, {'V': 120.0, 'C': 1e-4})])
def test_own_model_files_reproduce_reference_golden(clean_modules, golden, name, fixture, ticks, tol, scales):
    """tests/models/four_variable.py and eight_variable.py (our own transcriptions of the two published models)
    -> graph -> interpreter reproduce the golden trajectories generated from the reference's own files"""
    sys.path.insert(0, HERE)
    from traced_cases import make_model
    from oracle.graph_eval import Interpreter
    f = golden(fixture)
    H, W = f['phase'].shape
    m = make_model(name, H, W, diff=float(f['diff']))
    m.phase = f['phase']
    m.define()
    c = m._analyze()
    it = Interpreter(c, m.phase)
    st = state_of(f, m.VAR_NAMES)
    t0 = 0
    for t in ticks:
        st = it.tick(st, t - t0)
        t0 = t
        check(st, f, m.VAR_NAMES, t, tol, scales)


def test_conv_laplacian_model_reproduces_fenton_simple_golden(clean_modules, golden):
    """tests/models/simple_conv.py (tf.pad boundary + zero-padded 3x3 tf.nn.depthwise_conv2d, our own file in the style
    of the reference's stand-alone scripts) -> graph -> interpreter reproduces the golden trajectory generated from the
    reference's own fenton_simple.py; the generated source marks the model ZEROPAD and keeps one sub-step per launch"""
    sys.path.insert(0, HERE)
    from traced_cases import make_model
    from oracle.graph_eval import Interpreter
    import fib_tf_amd.tfgraph as tf
    from fib_tf_amd.traced import TraceError
    f = golden('fenton_simple_traj')
    H, W = f['init_U'].shape
    m = make_model('fvc', H, W, diff=float(f['diff']))
    m.define()
    c = m._analyze()
    assert c['spt'] == 1 and c['programs'][0][1].zeropad and c['programs'][0][1].uses_lap
    assert 'static constexpr bool ZEROPAD = true;' in c['source'] and '#define FIB_CUSTOM_K 1' in c['source']
    it = Interpreter(c, None)
    st = state_of(f, m.VAR_NAMES)
    t0 = 0
    for t in [1, 2, 10, 100]:                               # (the script's S2 fires at step 150)
        st = it.tick(st, t - t0)
        t0 = t
        check(st, f, m.VAR_NAMES, t, 1e-6)
    # what the tracer does not take for a stencil it refuses by name
    x = tf.Variable(np.zeros((8, 8)))
    with pytest.raises(NotImplementedError, match='no-flux boundary'):
        tf.pad(x, [[1, 1], [1, 1]], 'REFLECT')
    with pytest.raises(NotImplementedError, match='3x3 single-channel'):
        tf.nn.depthwise_conv2d(tf.expand_dims(tf.expand_dims(x, 0), -1), np.ones((5, 5, 1, 1)), [1, 1, 1, 1], 'SAME')
    bad = tf.nn.depthwise_conv2d(tf.expand_dims(tf.expand_dims(tf.pad(x[1:-1, 1:-1], [[1, 1], [1, 1]], 'SYMMETRIC'), 0), -1),
                                 np.ones((3, 3, 1, 1)), [1, 1, 1, 1], 'SAME')[0, :, :, 0]
    from fib_tf_amd.traced import IonicModel

    class M(IonicModel):
        def define(self):
            super().define()
            v = tf.Variable(np.zeros([16, 16], np.float32))
            self._ode_op = tf.group(v.assign(self.solve((v,))[0]))
    M.solve = lambda self, s: (s[0] + tf.nn.depthwise_conv2d(
        tf.expand_dims(tf.expand_dims(tf.pad(s[0][1:-1, 1:-1], [[1, 1], [1, 1]], 'SYMMETRIC'), 0), -1),
        np.ones((3, 3, 1, 1)), [1, 1, 1, 1], 'SAME')[0, :, :, 0],)
    mm = M({'height': 16, 'width': 16, 'dt': 0.1, 'diff': 1.0, 'duration': 1, 'dt_per_plot': 1})
    mm.define()
    with pytest.raises(TraceError, match='only 3x3 kernel'):
        mm.generated_source()
    del bad


@pytest.mark.parametrize('name,spt,kinds', [('ap', 10, 1), ('ms', 5, 1), ('gated', 1, 1), ('mrfhn', 4, 2)])
def test_test_models_trace_and_interpret(clean_modules, name, spt, kinds):
    sys.path.insert(0, HERE)
    from traced_cases import interpret, make_model
    m = make_model(name, 24, 32, (16, 12, 4))
    m.define()
    c = m._analyze()
    assert c['spt'] == spt == m.dt_per_step
    assert len({lv.signature() for lv in c['programs'][0][1].levels}) == kinds
    src = m.generated_source()
    assert src.count('struct Custom') == 1 and ('#define FIB_CUSTOM_K %d' % spt) in src
    assert m.VAR_NAMES[0] in ('u', 'v', 'V')               # the diffusing variable is slab entry 0
    st, trend = interpret(m, name, 12, 5)
    assert np.isfinite(st).all() and st.shape == (len(m.VAR_NAMES), 24, 32)
    if name == 'gated':
        assert [n for n, _ in c['programs']] == ['_ode_op', 'slow'] and trend.shape == (2, 2)
        assert 'MODE == 1' in src


def test_tracer_refuses_what_it_cannot_compile(clean_modules):
    import fib_tf_amd.tfgraph as tf
    from fib_tf_amd.traced import IonicModel, TraceError
    cfg0 = {'height': 16, 'width': 16, 'dt': 0.1, 'diff': 1.0, 'duration': 1, 'dt_per_plot': 1}
    z = np.zeros([16, 16], np.float32)

    def model(solve, n=1, post=None):
        class M(IonicModel):
            def define(self):
                super().define()
                vs = tuple(tf.Variable(z) for _ in range(n))
                out = self.solve(vs)
                if post:
                    out = post(out)
                self._ode_op = tf.group(*[v.assign(o) for v, o in zip(vs, out)])
        M.solve = solve
        m = M(dict(cfg0))
        m.define()
        return m

    with pytest.raises(TraceError, match='enforce_boundary'):
        model(lambda self, s: (s[0] + self.laplace(s[0]),)).generated_source()
    with pytest.raises(TraceError, match='only one variable may diffuse'):
        model(lambda self, s: (s[0] + self.laplace(self.enforce_boundary(s[0])),
                               s[1] + self.laplace(self.enforce_boundary(s[1]))), n=2).generated_source()
    with pytest.raises(TraceError, match='outside solve'):
        model(lambda self, s: (s[0] * 2.0,), post=lambda o: (o[0] + 1.0,)).generated_source()
    with pytest.raises(NotImplementedError, match='IonicModel.enforce_boundary'):
        tf.pad(None, None)                    # (the one pad that IS understood: test_conv_laplacian_model_...)
    m = model(lambda self, s: (s[0] * 2.0,))
    m._ode_op = None
    with pytest.raises(TraceError, match='_ode_op'):
        m.generated_source()
    # dt_per_step must agree with the chain define() built
    m = model(lambda self, s: (s[0] * 2.0,))
    m.dt_per_step = 7
    with pytest.raises(TraceError, match='dt_per_step'):
        m.generated_source()
