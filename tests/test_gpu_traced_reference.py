"""-m gpu: the reference's UNCHANGED model files, traced and compiled in the build container
(oracle/build_ref_traced.py -> oracle/_ref/traced/*.so, binaries that travel like oracle/_ref/generate_table),
run here through the C ABI against

  * the golden trajectories generated from the same files by the float32 stand-in (tests/golden), and
  * the hand-written kernels of libfibhip.so under the rounding-faithful policy — BITWISE: generated code and
    hand-written code implement the same graph with one float32 rounding per node.

Skipped when the binaries are absent (a checkout that never saw /root/reference)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACED = os.path.join(ROOT, 'oracle', '_ref', 'traced')


def load_case(case):
    so = os.path.join(TRACED, case + '.so')
    if not os.path.exists(so):
        pytest.skip('oracle/_ref/traced/%s.so was not built (needs the reference tree at build time)' % case)
    from fib_tf_amd import _lib
    with open(os.path.join(TRACED, case + '.json')) as f:
        meta = json.load(f)
    return _lib.load(so), meta


def traced_stepper(case, H, W, fast, phase, init, names):
    from fib_tf_amd import _lib
    L, meta = load_case(case)
    assert meta['names'] == list(names)
    st = _lib.Stepper(_lib.CUSTOM, H, W, meta['dt'], meta['diff'], flags=_lib.FAST if fast else 0, library=L)
    assert st.steps_per_tick == meta['spt']
    if phase is not None:
        st.set_phase(phase)
    st.set_state(-1, init)
    return st, meta


def close(got, want, tol, what, scale=None):
    s = scale if scale is not None else max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= tol * s, '%s: max|d| %.3e > %.1e*%g' % (what, err, tol, s)


@pytest.mark.parametrize('fixture,case', [('fenton_traj64', 'fenton_d1.5'), ('fenton_traj_ragged', 'fenton_d1.1')])
def test_reference_fenton_file_on_gpu(gpu_lib, golden, fixture, case):
    from fib_tf_amd import _lib
    f = golden(fixture)
    names = ('U', 'V', 'W', 'S')
    init = np.stack([f['init_' + n] for n in names])
    _, H, W = init.shape
    phase = f['phase'] if f['phase'].size else None
    st, meta = traced_stepper(case, H, W, False, phase, init, names)
    assert st.launch_plan() == (10, 1)                     # the ten chained solve() calls: one fused launch
    nat = _lib.Stepper(_lib.FENTON4V, H, W, 0.1, meta['diff'])
    if phase is not None:
        nat.set_phase(phase)
    nat.set_state(-1, init)
    t0 = 0
    for t in [int(x) for x in f['snap_ticks']]:
        st.step(t - t0)
        nat.step(t - t0)
        t0 = t
        got = st.get_state(-1)
        for i, n in enumerate(names):
            close(got[i], f['%s_t%d' % (n, t)], 2e-5 if t <= 20 else 3e-4, '%s %s t%d' % (case, n, t), 1.0)
        assert np.array_equal(got, nat.get_state(-1)), 'generated vs hand-written kernel differ at tick %d' % t
    # hardware-rate policy of the generated code: same trajectory within the fast tolerance
    fs, _ = traced_stepper(case, H, W, True, phase, init, names)
    fs.step(20)
    g = fs.get_state(-1)
    for i, n in enumerate(names):
        close(g[i], f['%s_t20' % n], 2e-4, '%s fast %s t20' % (case, n), 1.0)


BR_NAMES = ('V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI')


@pytest.mark.parametrize('fixture,case,cheby,skip', [('br_traj64_cheby', 'br_cheby_d0.809', True, False),
                                                     ('br_traj64_direct', 'br_direct_d0.809', False, False),
                                                     ('br_traj64_skip', 'br_skip_d0.809', False, True)])
def test_reference_br_file_on_gpu(gpu_lib, golden, fixture, case, cheby, skip):
    from fib_tf_amd import _lib
    from fib_tf_amd.br import BeelerReuter
    f = golden(fixture)
    init = np.stack([f['init_' + n] for n in BR_NAMES])
    _, H, W = init.shape
    st, meta = traced_stepper(case, H, W, False, f['phase'], init, BR_NAMES)
    assert st.launch_plan() == (5, 1)
    nat = _lib.Stepper(_lib.BR, H, W, 0.1, meta['diff'],
                       flags=(_lib.CHEBY if cheby else 0) | (_lib.SKIP if skip else 0))
    if cheby:
        nat.set_consts(BeelerReuter({'height': 8, 'width': 8}).chebyshev_table())
    nat.set_phase(f['phase'])
    nat.set_state(-1, init)
    scales = {'V': 120.0, 'C': 1e-4}
    t0 = 0
    for t in [int(x) for x in f['snap_ticks']]:
        st.step(t - t0)
        nat.step(t - t0)
        t0 = t
        got, hand = st.get_state(-1), nat.get_state(-1)
        for i, n in enumerate(BR_NAMES):
            close(got[i], f['%s_t%d' % (n, t)], 3e-5, '%s %s t%d' % (case, n, t), scales.get(n, 1.0))
            # the hand-written kernel folds a few constants differently (e.g. exp(0*x) rows of the rate table),
            # so equality with it is to rounding, not bitwise
            close(got[i], hand[i], 3e-5, '%s vs hand-written %s t%d' % (case, n, t), scales.get(n, 1.0))


def test_reference_court_file_on_gpu(gpu_lib, golden):
    f = golden('court_traj64')
    names = [str(n) for n in f['names']]
    init = np.stack([f['init_' + n] for n in names])
    _, H, W = init.shape
    st, meta = traced_stepper('court_d0.809', H, W, False, f['phase'], init, names)
    assert [m['name'] for m in meta['modes']] == ['_ode_op', 'slow'] and len(meta['modes'][1]['mask']) == 17
    scales = {'V': 150.0, '_Ca_i_': 1e-3, '_Ca_rel_': 1.5, '_Ca_up_': 1.5, '_K_i_': 139.0, '_Na_i_': 11.0}
    t0 = 0
    for t in [1, 2, 10, 11, 100]:
        for i in range(t0, t):                             # court.py:612-617
            st.step(1)
            if i % 10 == 0:
                st.step_mode(1)
        t0 = t
        got = st.get_state(-1)
        for i, n in enumerate(names):
            close(got[i], f['%s_t%d' % (n, t)], 3e-5, 'court %s t%d' % (n, t), scales.get(n, 1.0))
