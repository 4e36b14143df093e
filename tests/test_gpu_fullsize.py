"""-m gpu: parity at BASELINE.json's FULL sizes (configs[1], [2], [4]) and size-independent properties.

The oracle runs the same workload on the host cores (a second or two each); beyond that, the checks use
properties that hold at any size: a launch plan must never change a bit of the result (fusion depth,
tile shape, kernel family), and a mirror-symmetric problem must stay mirror-symmetric.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = {'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000, 'skip': False, 'cheby': True}


def state(m):
    return np.stack([m._State[n].eval() for n in m.VAR_NAMES])


def advance(m, ticks, hook=None):
    m.duration = ticks * m.dt_per_step * m.dt + 1e-9
    for i in m.run():
        if hook:
            hook(i)


# Tolerances of the full-size oracle comparisons: at most ten times what the GPU measures on these very runs (every test prints
# its measured figures; DESIGN.md 3 lists them), so that a regression of the kernels that run at BASELINE's sizes — the
# multi-tick launches, their stop / run-ahead paths — is caught HERE, where they meet the checker, and not only by the
# bit-identity tests against one launch per tick.  The runs include the S2 stimulus: the broken wave it leaves amplifies rounding
# differences far more than the plane S1 wave does (a run without S2 ends 1000 sub-steps at 4e-7), and under the fast policy a few
# cells sit on the other side of one of the model's Heaviside switches for a step (fenton.py:73-79,87: V jumps by dt*V/tau) —
# hence a bound on the bulk (99.99th percentile) next to the bound on the worst cell there.
FENTON_500_TOL = {'fast': 3e-5, 'exact': 5e-6}            # 200 sub-steps after S2.  measured: 3.0e-6 / 5.1e-7
FENTON_1000_TOL = {'fast': 0.1, 'exact': 4e-4}            # worst cell, 700 sub-steps after S2.  measured: 2.3e-2 / 4.1e-5
FENTON_1000_BULK = {'fast': 1.5e-3, 'exact': 2e-4}        # 99.99th percentile.  measured: 1.5e-4 / 2.1e-5
BR_100_TOL_MV = {'fast': 4e-3, 'exact': 2.5e-4}           # 100 sub-steps with S2.  measured: 7.7e-4 mV / 2.3e-5 mV
BR_100_TOL_GATE = {'fast': 1.6e-4, 'exact': 5e-6}         # measured: 1.6e-5 / 4.8e-7
COURT_21_TOL_MV = {'fast': 2e-3, 'exact': 2e-3}           # 21 ticks with three 'slow' ops.  measured: 2.0e-4 mV / 2.1e-4 mV
COURT_21_TOL_REL = {'fast': 1e-4, 'exact': 1e-4}          # worst array (the SR release gate u, court.py:243): 1.5e-5 of its range


def pace_rect_luq(H, W):
    return 1, H // 2, 1, W // 2                           # ionic.py:152


def err_report(got, ref, tol):
    """(max |d|, values beyond tol, values beyond 10 tol, text) — the text goes into every assertion message"""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    n1, n10 = int((d > tol).sum()), int((d > 10 * tol).sum())
    return float(d.max()), n1, n10, 'max %.3g, %d of %d values beyond %.1g, %d beyond %.1g, 99.99th percentile %.3g' % (
        d.max(), n1, d.size, tol, n10, 10 * tol, np.percentile(d, 99.99))


@pytest.mark.parametrize('policy', ['fast', 'exact'])
def test_fenton_512_config1_vs_oracle(gpu_lib, orc, policy):
    """fenton.py __main__ (fenton.py:156-187): 512x512, diff 1.5, hole (256,256,30), S1, S2 'luq' fired in mid-run, a frame
    read back in mid-run (image(): the run-ahead / stop path of the multi-tick launches); 1000 sub-steps"""
    from fib_tf_amd.fenton import Fenton4v
    m = Fenton4v(dict(BASE, height=512, width=512, diff=1.5, fast_math=policy == 'fast'))
    m.add_hole_to_phase_field(256, 256, 30)
    m.define()
    m.add_pace_op('s2', 'luq', 1.0)
    ref = state(m)
    frames = {}

    def hook(i):
        if i == 29:
            m.fire_op('s2')
        if i in (49, 59, 69):
            frames[i] = m.image().copy()
    advance(m, 100, hook)
    assert m._stepper.ticks_per_launch() > 1              # this IS the multi-tick path
    # the oracle: the same ticks, S2 after tick 29 (fire_op follows the tick's sess.run, fenton.py:181-183)
    orc.fenton_run(ref, 0.1, 1.5, m.phase, 300)
    r0, r1, c0, c1 = pace_rect_luq(512, 512)
    ref[0] = orc.pace(ref[0], r0, r1, c0, c1, 1.0, 0.0)
    orc.fenton_run(ref, 0.1, 1.5, m.phase, 200)
    mid = ref[0].copy()
    orc.fenton_run(ref, 0.1, 1.5, m.phase, 500)
    got = state(m)
    err, n1, n10, text = err_report(got, ref, FENTON_1000_BULK[policy])
    err_mid, m1, m10, text_mid = err_report(frames[49], mid, FENTON_500_TOL[policy])
    bulk = float(np.percentile(np.abs(got.astype(np.float64) - ref), 99.99))
    print('fenton 512 %s: final state %s; frame after 500 sub-steps %s' % (policy, text, text_mid))
    assert err_mid <= FENTON_500_TOL[policy], 'frame after 500 sub-steps: ' + text_mid
    assert bulk <= FENTON_1000_BULK[policy], 'after 1000 sub-steps: ' + text
    assert err <= FENTON_1000_TOL[policy], 'after 1000 sub-steps: ' + text
    assert got[0].max() > 0.9 and frames[69][r0 + 5, c0 + 5] > 0.5            # the S2 stimulus is really there


@pytest.mark.parametrize('policy', ['fast', 'exact'])
def test_fenton_512_plan_invariance(gpu_lib, policy, monkeypatch):
    """the default plan (10 fused sub-steps, strip kernel), 2 x 5 fused, and 10 single-step launches give
    bit-identical states at full size, pacing included"""
    from fib_tf_amd.fenton import Fenton4v
    out = {}
    for variant in ('', '5,54,21,-3', '5,32,32,256', '1,64,4,256'):
        if variant:
            monkeypatch.setenv('FIBHIP_VARIANT', variant)
        else:
            monkeypatch.delenv('FIBHIP_VARIANT', raising=False)
        m = Fenton4v(dict(BASE, height=512, width=512, diff=1.5, fast_math=policy == 'fast'))
        m.add_hole_to_phase_field(256, 256, 30)
        m.define()
        m.add_pace_op('s2', 'luq', 1.0)
        advance(m, 30, lambda i: m.fire_op('s2') if i == 21 else None)
        out[variant] = state(m)
    for k, v in out.items():
        assert np.array_equal(v, out['']), 'plan %r changes the result' % k


@pytest.mark.parametrize('size,variants', [(768, ('5,54,23,-3', '10,44,28,-3', '1,64,4,256')),
                                           (1536, ('5,54,22,-4', '5,54,21,-3'))])
def test_fenton_plan_invariance_other_sizes(gpu_lib, size, variants, monkeypatch):
    """the size-dependent default plans (wave-exact 54x23 / 54x22 tiles, fibhip.hip build_plan) against other
    shapes: bit-identical"""
    from fib_tf_amd.fenton import Fenton4v
    out = {}
    for variant in ('',) + variants:
        if variant:
            monkeypatch.setenv('FIBHIP_VARIANT', variant)
        else:
            monkeypatch.delenv('FIBHIP_VARIANT', raising=False)
        m = Fenton4v(dict(BASE, height=size, width=size, diff=1.5))
        m.add_hole_to_phase_field(size // 2, size // 2, size // 17)
        m.define()
        m.add_pace_op('s2', 'luq', 1.0)
        advance(m, 8, lambda i: m.fire_op('s2') if i == 3 else None)
        out[variant] = state(m)
    for k, v in out.items():
        assert np.array_equal(v, out['']), 'plan %r changes the result at %d^2' % (k, size)


def test_fenton_4096_config3_grid_vs_oracle(gpu_lib, orc, monkeypatch):
    """BASELINE configs[3]'s 4096x4096 grid on ONE GPU (hole (2048,2048,240), S1): 20 sub-steps against the oracle,
    and the default plan (5 fused sub-steps, 54x22 tiles) bit-identical to one sub-step per launch"""
    from fib_tf_amd.fenton import Fenton4v
    res = []
    for k in ('', '1'):
        if k:
            monkeypatch.setenv('FIBHIP_K', k)
        else:
            monkeypatch.delenv('FIBHIP_K', raising=False)
        m = Fenton4v(dict(BASE, height=4096, width=4096, diff=1.5))
        m.add_hole_to_phase_field(2048, 2048, 240)
        m.define()
        if not k:
            ref = state(m)
            assert m._stepper.launch_plan() == (5, 2)
        advance(m, 2)
        res.append(state(m))
        m._stepper.close()
    assert np.array_equal(res[0], res[1])
    orc.fenton_run(ref, 0.1, 1.5, m.phase, 20)
    err = np.abs(res[0].astype(np.float64) - ref).max()
    assert err <= 2e-5, err


def test_fenton_mirror_symmetry_1024(gpu_lib):
    """a problem that is mirror-symmetric about the horizontal mid-line stays so (to round-off: the
    reference's diagonal sum is not associative under the mirror)"""
    from fib_tf_amd.fenton import Fenton4v
    H = 1024
    m = Fenton4v(dict(BASE, height=H, width=640, diff=1.5))
    m.add_hole_to_phase_field(300, (H - 1) / 2.0, 40)
    assert np.array_equal(m.phase, m.phase[::-1])
    m.define()
    advance(m, 40)
    s = state(m)
    assert np.abs(s - s[:, ::-1]).max() < 2e-5
    assert s[0].max() > 0.9                       # the S1 wave is really travelling


@pytest.mark.parametrize('policy', ['fast', 'exact'])
def test_br_512_config2_vs_oracle(gpu_lib, orc, policy):
    """br.py __main__ (br.py:348-365): 512x512, diff 0.809, cheby=True, hole (150,200,40), S2 'luq' and a read-back in
    mid-run; 100 sub-steps"""
    from fib_tf_amd.br import BeelerReuter
    m = BeelerReuter(dict(BASE, height=512, width=512, diff=0.809, fast_math=policy == 'fast'))
    m.add_hole_to_phase_field(150, 200, 40)
    m.define()
    m.add_pace_op('s2', 'luq', 10.0)
    ref = state(m)
    frames = {}

    def hook(i):
        if i == 7:
            m.fire_op('s2')
        if i in (11, 15):
            frames[i] = m.pot().eval().copy()
    advance(m, 20, hook)
    assert m._stepper.ticks_per_launch() > 1
    tbl = m.chebyshev_table().astype(np.float32)
    orc.br_run(ref, 0.1, 0.809, m.phase, tbl, False, 8)
    r0, r1, c0, c1 = pace_rect_luq(512, 512)
    ref[0] = orc.pace(ref[0], r0, r1, c0, c1, 10.0, -90.0)
    orc.br_run(ref, 0.1, 0.809, m.phase, tbl, False, 4)
    mid = ref[0].copy()
    orc.br_run(ref, 0.1, 0.809, m.phase, tbl, False, 8)
    got = state(m)
    ev = np.abs(got[0].astype(np.float64) - ref[0]).max()
    eg = np.abs(got[2:].astype(np.float64) - ref[2:]).max()
    em = np.abs(frames[11].astype(np.float64) - mid).max()
    print('br 512 %s: |dV| %.3g mV after 100 sub-steps, %.3g mV after 60, gates %.3g' % (policy, ev, em, eg))
    assert ev <= BR_100_TOL_MV[policy], 'max |dV| after 100 sub-steps: %.3g mV (tolerance %.1g)' % (ev, BR_100_TOL_MV[policy])
    assert em <= BR_100_TOL_MV[policy], 'frame after 60 sub-steps: %.3g mV' % em
    assert eg <= BR_100_TOL_GATE[policy], 'gates: %.3g (tolerance %.1g)' % (eg, BR_100_TOL_GATE[policy])
    assert got[0][r0 + 5, c0 + 5] > -20.0                 # the stimulated quadrant is depolarised


@pytest.mark.parametrize('policy', ['fast', 'exact'])
def test_court_1024_config4_vs_oracle(gpu_lib, orc, policy):
    """court.py protocol scaled x2: 1024x1024, two holes, fast tick + 'slow' every 10th"""
    from fib_tf_amd.court import Courtemanche
    m = Courtemanche(dict(BASE, height=1024, width=1024, diff=0.809, fast_math=policy == 'fast'))
    m.add_hole_to_phase_field(512, 512, 60)
    m.add_hole_to_phase_field(512, 512, 500, neg=True)
    m.define()
    ref = state(m)
    advance(m, 21, lambda i: m.fire_op('slow') if i % 10 == 0 else None)
    orc.court_run(ref, 0.1, 0.809, m.phase, True, 0, 21)
    got = state(m)
    ev = np.abs(got[0].astype(np.float64) - ref[0]).max()
    assert ev <= COURT_21_TOL_MV[policy], 'max |dV| after 21 ticks: %.3g mV (tolerance %.1g)' % (ev, COURT_21_TOL_MV[policy])
    worst = (0.0, '')
    for i in range(1, 21):
        sc = max(float(np.abs(ref[i]).max()), 1e-3)
        worst = max(worst, (float(np.abs(got[i].astype(np.float64) - ref[i]).max()) / sc, m.VAR_NAMES[i]))
    print('court 1024 %s: |dV| %.3g mV after 21 ticks, worst array %s %.3g of its range' % (policy, ev, worst[1], worst[0]))
    assert worst[0] <= COURT_21_TOL_REL[policy], 'worst array %s: %.3g of its range (tolerance %.1g)' % (worst[1], worst[0], COURT_21_TOL_REL[policy])


def test_court_1024_ticks_per_launch_invariance(gpu_lib, monkeypatch):
    """configs[4] at full size under the default policy: three ticks per launch (the shape picked by measurement on this
    grid), one tick per launch (FIBHIP_NO_MULTI) and the other three-tick shapes of the table leave the same bits — with
    'slow' every 10th tick, a pace and a read-back falling between the ticks"""
    from fib_tf_amd.court import Courtemanche
    out = {}
    for name, env in (('default', {}), ('one tick per launch', {'FIBHIP_NO_MULTI': '1'}),
                      ('58x28 strips', {'FIBHIP_COURT_MULTI3': '58,28,-2'}), ('32x32 tiles', {'FIBHIP_COURT_MULTI3': '32,32,256'})):
        for k in ('FIBHIP_NO_MULTI', 'FIBHIP_COURT_MULTI3'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = Courtemanche(dict(BASE, height=1024, width=1024, diff=0.809))
        m.add_hole_to_phase_field(512, 512, 60)
        m.define()
        m.add_pace_op('s2', 'luq', 10.0)
        if name == 'default':
            assert m._stepper.ticks_per_launch() == 3

        def hook(i):
            if i % 10 == 0:
                m.fire_op('slow')
            if i == 13:
                m.fire_op('s2')
            if i == 17:
                m._State['_m_'].eval()

        advance(m, 26, hook)
        out[name] = state(m)
    for k, v in out.items():
        assert np.array_equal(v, out['default']), '%s changes the result' % k
