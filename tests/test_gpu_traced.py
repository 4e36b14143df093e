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
    assert 'libfibhip_' in m._library._name and '_traced' in m._library._name
    fused, launches = m._stepper.launch_plan()
    assert (fused, launches) == (m.dt_per_step, 1)
    assert m.generated_source().count('struct Custom') == 1
    del ctypes


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
