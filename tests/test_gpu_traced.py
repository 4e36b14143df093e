"""-m gpu: models written in the reference's style (tests/models/*.py), traced by fib_tf_amd.traced, compiled to
HIP and run through the C ABI — against the op-by-op interpreter of the same graph (oracle/graph_eval.py, itself
pinned to the reference's own model files by tests/test_traced_cpu.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traced_cases import MODELS, drive, interpret, make_model  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [('ap', 45, 70, (30, 20, 6), 30, 12), ('ms', 64, 48, (20, 30, 5), 40, 15),
         ('gated', 40, 56, (28, 20, 5), 150, 60), ('mrfhn', 50, 50, None, 40, 15)]


@pytest.fixture(autouse=True)
def _restore_modules():
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'ionic', 'screen')}
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


@pytest.mark.parametrize('policy', ['exact', 'fast'])
@pytest.mark.parametrize('name,H,W,hole,ticks,s2', CASES, ids=[c[0] for c in CASES])
def test_traced_model_matches_graph_interpreter(gpu_lib, name, H, W, hole, ticks, s2, policy):
    m = make_model(name, H, W, hole, fast_math=(policy == 'fast'))
    m.define()
    got, trend = drive(m, name, ticks, s2)
    ref = make_model(name, H, W, hole)
    ref.define()
    want, wtrend = interpret(ref, name, ticks, s2)
    assert m.VAR_NAMES == ref.VAR_NAMES
    # exact: one float32 rounding per graph node on both sides, transcendental functions a few ulp apart (ocml vs
    # NumPy) and amplified by the wavefront; fast: hardware exp/rcp forms
    tol = 2e-5 if policy == 'exact' else 2e-3
    for i, n in enumerate(m.VAR_NAMES):
        scale = max(1.0, float(np.abs(want[i]).max()))
        err = float(np.abs(got[i] - want[i]).max())
        assert err <= tol * scale, '%s %s: max|d| %.3e > %.1e*%g' % (name, n, err, tol, scale)
    if name == 'gated':
        assert trend.shape == wtrend.shape == (ticks // 10, 2)
        assert np.allclose(trend, wtrend, rtol=0, atol=tol * 130)
    # the library that ran is the one generated for this model, and the whole tick is one fused launch
    import ctypes
    # ... compiled in this process (hiprtc) and loaded into the stock library as a run-time module
    assert '#module:model_' in m._library._name and m._library.module.value
    fused, launches = m._stepper.launch_plan()
    assert (fused, launches) == (m.dt_per_step, 1)
    assert m.generated_source().count('struct Custom') == 1
    del ctypes


def _golden_init(f, names):
    return np.stack([f['init_' + n] for n in names])


def _close(got, want, tol, what, scale=None):
    s = scale if scale is not None else max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= tol * s, '%s: max|d| %.3e > %.1e*%g' % (what, err, tol, s)


@pytest.mark.parametrize('fixture', ['fenton_traj64', 'fenton_traj_ragged'])
def test_generated_four_variable_equals_handwritten(gpu_lib, golden, fixture):
    """tests/models/four_variable.py (our own model file in the reference's style) -> tracer -> generated kernel,
    against (a) the golden trajectory the reference's own fenton.py produced and (b) the hand-written Fenton
    kernel under the rounding-faithful policy: BITWISE — generated and hand-written code state the same graph with
    one float32 rounding per node"""
    from fib_tf_amd import _lib
    f = golden(fixture)
    names = ('U', 'V', 'W', 'S')
    init = _golden_init(f, names)
    _, H, W = init.shape
    phase = f['phase'] if f['phase'].size else None
    for fast, tol20 in ((False, 2e-5), (True, 2e-4)):
        m = make_model('fv', H, W, fast_math=fast, diff=float(f['diff']))
        m.phase = phase
        m.define()
        m._ensure_compiled()
        assert m.VAR_NAMES == names and m._stepper.launch_plan() == (10, 1)
        m._stepper.set_state(-1, init)
        nat = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, float(f['diff']), flags=_lib.FAST if fast else 0)
        if phase is not None:
            nat.set_phase(phase)
        nat.set_state(-1, init)
        t0 = 0
        for t in [int(x) for x in f['snap_ticks']]:
            if fast and t > 20:
                break
            m._stepper.step(t - t0)
            nat.step(t - t0)
            t0 = t
            got = m._stepper.get_state(-1)
            for i, n in enumerate(names):
                _close(got[i], f['%s_t%d' % (n, t)], tol20 if t <= 20 else 3e-4, 'fv %s t%d' % (n, t), 1.0)
            if not fast:
                assert np.array_equal(got, nat.get_state(-1)), 'generated vs hand-written kernel differ at tick %d' % t


def test_generated_conv_variant_equals_handwritten(gpu_lib, golden):
    """tests/models/simple_conv.py — the boundary as `tf.pad`, the Laplacian as a zero-padded 3x3
    `tf.nn.depthwise_conv2d` — -> generated kernel (ZEROPAD, one sub-step per launch), against the golden trajectory
    of the reference's fenton_simple.py and, under the rounding-faithful policy, BITWISE against the hand-written
    FIBHIP_ZEROPAD kernel"""
    from fib_tf_amd import _lib
    f = golden('fenton_simple_traj')
    names = ('U', 'V', 'W', 'S')
    init = _golden_init(f, names)
    _, H, W = init.shape
    for fast in (False, True):
        m = make_model('fvc', H, W, fast_math=fast, diff=float(f['diff']))
        m.define()
        m._ensure_compiled()
        assert m.VAR_NAMES == names and m._stepper.launch_plan() == (1, 1)
        assert 'ZEROPAD = true' in m.generated_source()
        m._stepper.set_state(-1, init)
        nat = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, float(f['diff']), flags=_lib.ZEROPAD | (_lib.FAST if fast else 0),
                           steps_per_tick=1)
        nat.set_state(-1, init)
        t0 = 0
        for t in [1, 2, 10, 100]:                           # (the script's own S2 fires at step 150)
            m._stepper.step(t - t0)
            nat.step(t - t0)
            t0 = t
            got = m._stepper.get_state(-1)
            for i, n in enumerate(names):
                _close(got[i], f['%s_t%d' % (n, t)], 2e-5 if not fast else 2e-4, 'fvc %s t%d' % (n, t), 1.0)
            if not fast:
                assert np.array_equal(got, nat.get_state(-1)), 'generated vs hand-written ZEROPAD kernel differ at step %d' % t


def test_generated_eight_variable_vs_handwritten(gpu_lib, golden):
    """tests/models/eight_variable.py -> generated kernel, against the golden trajectory of the reference's br.py
    (direct gates) and the hand-written Beeler-Reuter kernel.  The hand-written kernel folds a few constants
    differently (the exp(0*x) rows of the rate table), so equality with it is to rounding, not bitwise."""
    from fib_tf_amd import _lib
    f = golden('br_traj64_direct')
    names = ('V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI')
    init = _golden_init(f, names)
    _, H, W = init.shape
    m = make_model('ev', H, W, fast_math=False, diff=float(f['diff']))
    m.phase = f['phase']
    m.define()
    m._ensure_compiled()
    assert m.VAR_NAMES == names and m._stepper.launch_plan() == (5, 1)
    m._stepper.set_state(-1, init)
    nat = _lib.Stepper(_lib.BR, H, W, 0.1, float(f['diff']))
    nat.set_phase(f['phase'])
    nat.set_state(-1, init)
    scales = {'V': 120.0, 'C': 1e-4}
    t0 = 0
    for t in [int(x) for x in f['snap_ticks']]:
        m._stepper.step(t - t0)
        nat.step(t - t0)
        t0 = t
        got, hand = m._stepper.get_state(-1), nat.get_state(-1)
        for i, n in enumerate(names):
            _close(got[i], f['%s_t%d' % (n, t)], 3e-5, 'ev %s t%d' % (n, t), scales.get(n, 1.0))
            _close(got[i], hand[i], 3e-5, 'ev vs hand-written %s t%d' % (n, t), scales.get(n, 1.0))


@pytest.mark.parametrize('name,ticks', [('ap', 12), ('gated', 35), ('fvc', 20)])
def test_inprocess_build_equals_compiler_build(gpu_lib, name, ticks, monkeypatch):
    """the two ways a traced model reaches the GPU — hiprtc in this process -> code object -> fibhip_module_load into
    the stock library (default), and hipcc -> a per-model build of libfibhip (FIBTF_TRACED_BUILD=hipcc) — run the same
    generated source through the same kernels: bitwise equal states, incl. the 'slow' assign group of `gated`
    (pointwise_kernel through the module API) and the ZEROPAD variant"""
    res = []
    for how in ('hiprtc', 'hipcc'):
        monkeypatch.setenv('FIBTF_TRACED_BUILD', how)
        m = make_model(name, 60, 72, (30, 30, 7) if name != 'fvc' else None)
        m.define()
        st, trend = drive(m, name, ticks, 5)
        assert ('#module:' in m._library._name) == (how == 'hiprtc')
        res.append((st, trend, m._stepper.launch_plan()))
    assert res[0][2] == res[1][2]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize('name', ['ap', 'mrfhn'])
@pytest.mark.parametrize('policy', ['exact', 'fast'])
def test_fusion_depth_does_not_change_a_bit(gpu_lib, name, policy, monkeypatch):
    """the temporally blocked launch (all sub-steps of the tick in one kernel) and one launch per sub-step run
    the same generated arithmetic: bitwise equal states"""
    def run():
        m = make_model(name, 70, 90, (40, 30, 8), fast_math=(policy == 'fast'))
        m.define()
        st, _ = drive(m, name, 12, 5)
        return st, m._stepper.launch_plan()
    a, plan_a = run()
    monkeypatch.setenv('FIBHIP_K', '1')
    b, plan_b = run()
    assert plan_a[1] == 1 and plan_b == (1, plan_a[0])
    assert np.array_equal(a, b)



@pytest.mark.parametrize('name', ['ap', 'fv', 'ev', 'mrfhn'])
@pytest.mark.parametrize('policy', ['exact', 'fast'])
def test_generated_kernels_run_several_ticks_per_launch(gpu_lib, name, policy, monkeypatch):
    """a traced model's tick-fusing strip kernel also exists as the multi-tick launch (kind 3 of its run-time module):
    up to 32 ticks per launch with the tiles handing their rims to each other, on arrays of 2 (padded 16-byte cell), 4
    and 8 variables — bitwise equal to one launch per tick (FIBHIP_MT=0), with a pace and read-backs in between"""
    def run(mt):
        if mt:
            monkeypatch.delenv('FIBHIP_MT', raising=False)
        else:
            monkeypatch.setenv('FIBHIP_MT', '0')
        monkeypatch.setenv('FIBHIP_AUTOTUNE', '0')              # the header's own plan: the tick-fusing strip
        m = make_model(name, 96, 130, (40, 50, 9), fast_math=(policy == 'fast'))
        m.define()
        state, _ = drive(m, name, 23, 7)                        # run(): single-tick calls -> launches of 1, 2, 4, ... ticks
        st = m._stepper
        st.step(37)                                             # one call of many ticks: a 32-tick launch + the rest
        out = [state, st.get_state(0).copy()]
        st.step(5)
        out.append(st.get_state(-1))
        return out, st.ticks_per_launch(), st.launch_plan()
    (a, tpl, plan), (b, _, _) = run(True), run(False)
    if plan[1] == 1 and plan[0] > 1:
        assert tpl > 1, 'the multi-tick form of the generated strip kernel was not used (plan %r)' % (plan,)
    for x, y in zip(a, b):
        assert np.isfinite(x).all() and np.array_equal(x, y)


def test_traced_model_errors_are_loud(gpu_lib):
    """what the tracer cannot compile it refuses by name — there is no fallback path"""
    import fib_tf_amd.tfgraph as tf
    from fib_tf_amd.traced import IonicModel, TraceError

    class RawLaplace(IonicModel):
        def solve(self, state):
            (u,) = state
            return (u + self.dt * self.laplace(u),)        # no enforce_boundary

        def define(self):
            super().define()
            u = tf.Variable(np.zeros([self.height, self.width], np.float32))
            self._ode_op = tf.group(u.assign(self.solve((u,))[0]))

    m = RawLaplace({'height': 16, 'width': 16, 'dt': 0.1, 'diff': 1.0, 'duration': 1, 'dt_per_plot': 1})
    m.min_v = 0.0
    m.define()
    with pytest.raises(TraceError, match='enforce_boundary'):
        next(iter(m.run()))

    class TwoSpecies(IonicModel):
        def solve(self, state):
            u, v = state
            return (u + self.laplace(self.enforce_boundary(u)), v + self.laplace(self.enforce_boundary(v)))

        def define(self):
            super().define()
            z = np.zeros([self.height, self.width], np.float32)
            u, v = tf.Variable(z), tf.Variable(z)
            u1, v1 = self.solve((u, v))
            self._ode_op = tf.group(u.assign(u1), v.assign(v1))

    m = TwoSpecies({'height': 16, 'width': 16, 'dt': 0.1, 'diff': 1.0, 'duration': 1, 'dt_per_plot': 1})
    m.define()
    with pytest.raises(TraceError, match='only one variable may diffuse'):
        m.generated_source()
    with pytest.raises(TypeError, match='truth value'):
        bool(tf.Variable(np.zeros((4, 4))) > 0)
    with pytest.raises(NotImplementedError, match='IonicModel.enforce_boundary'):
        tf.pad(None, None)


def _random_model(seed, H=24, W=40):
    """a model whose solve() is a seeded random expression over the whole op list of fib_tf_amd.tfgraph (arguments
    kept inside every op's well-conditioned domain), three variables, three chained sub-steps"""
    import fib_tf_amd.tfgraph as tf
    from fib_tf_amd.traced import IonicModel
    rng = np.random.default_rng(seed)

    def expr(pool, depth):
        if depth == 0 or rng.random() < 0.15:
            return pool[rng.integers(len(pool))] if rng.random() < 0.8 else float(np.round(rng.uniform(-2, 2), 3))
        k = rng.integers(16)
        a, b = expr(pool, depth - 1), expr(pool, depth - 1)
        if not isinstance(a, tf.Tensor):
            a = pool[rng.integers(len(pool))]
        if k == 0: return a + b
        if k == 1: return a - b
        if k == 2: return a * b
        if k == 3: return a / (tf.abs(b) + 1.5) if isinstance(b, tf.Tensor) else a / (abs(b) + 1.5)
        if k == 4: return tf.tanh(a)
        if k == 5: return tf.exp(tf.clip_by_value(a, -3.0, 2.0))
        if k == 6: return tf.log(tf.abs(a) + 1.0)
        if k == 7: return tf.sqrt(tf.abs(a) + 0.25)
        if k == 8: return tf.where(a > b, a * 0.5, b) if isinstance(b, tf.Tensor) else tf.where(a > b, a * 0.5, 1.0 - a)
        if k == 9: return (1 + tf.sign(a - 0.1)) * 0.5
        if k == 10: return tf.maximum(a, b)
        if k == 11: return tf.minimum(a, 0.75)
        if k == 12: return tf.pow(a, 3) * 0.1
        if k == 13: return tf.reciprocal(tf.square(a) + 1.0)
        if k == 14: return tf.expm1(tf.clip_by_value(a, -2.0, 0.5) * (0.01 if rng.random() < 0.5 else 1.0))
        return -a

    class Random(IonicModel):
        def __init__(self, props):
            IonicModel.__init__(self, props)
            self.min_v, self.max_v, self.depol = -1.0, 1.0, 0.0

        def solve(self, state):
            u, v, w = state
            u0 = self.enforce_boundary(u)
            pool = [u, v, w, u0]
            du, dv, dw = (expr(pool, 6) for _ in range(3))
            u1 = tf.clip_by_value(u0 + self.dt * du + self.diff * self.dt * self.laplace(u0), -2.0, 2.0)
            v1 = tf.clip_by_value(v + self.dt * dv, -2.0, 2.0)
            w1 = self.rush_larsen(w, tf.reciprocal(1.0 + tf.exp(-u0)), 0.5 + tf.square(dw) * 0.1, self.dt)
            return u1, v1, w1

        def define(self):
            super().define()
            r = np.random.default_rng(seed + 1000)
            init = [r.uniform(-1, 1, (self.height, self.width)).astype(np.float32) for _ in range(3)]
            init[2] = np.abs(init[2])
            U, V, Wv = (tf.Variable(a, name=n) for a, n in zip(init, 'uvw'))
            st = (U, V, Wv)
            for _ in range(3):
                st = self.solve(st)
            self.dt_per_step = 3
            self._ode_op = tf.group(U.assign(st[0]), V.assign(st[1]), Wv.assign(st[2]))

    m = Random({'height': H, 'width': W, 'dt': 0.05, 'diff': 0.8, 'dt_per_plot': 1, 'duration': 1000})
    m.add_hole_to_phase_field(W // 2, H // 2, 5)
    return m


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_expression_graphs(gpu_lib, seed):
    """code-generator fuzz: random graphs over every supported op, GPU (rounding-faithful policy) against the
    op-by-op interpreter; the solve() body consumes the RNG identically on every trace, so two instances of the same
    seed build the same graph"""
    from oracle.graph_eval import Interpreter
    m = _random_model(seed)
    m.define()
    m.fast_math = False
    ref = _random_model(seed)
    ref.define()
    assert m.generated_source() == ref.generated_source()
    c = ref._analyze()
    want = Interpreter(c, ref.phase).tick(np.stack([v.init for v in c['slots']]), 4)
    m.duration = 4 * m.dt_per_step * m.dt + 1e-9
    for _ in m.run():
        pass
    got = np.stack([m._State[n].eval() for n in m.VAR_NAMES])
    assert np.isfinite(want).all() and np.isfinite(got).all()
    for i, n in enumerate(m.VAR_NAMES):
        err = float(np.abs(got[i] - want[i]).max())
        assert err <= 3e-5, 'seed %d %s: max|d| %.3e' % (seed, n, err)
