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


def test_fenton_512_config1_vs_oracle(gpu_lib, orc):
    """fenton.py __main__ (fenton.py:156-171): 512x512, diff 1.5, hole (256,256,30), S1; 1000 sub-steps"""
    from fib_tf_amd.fenton import Fenton4v
    m = Fenton4v(dict(BASE, height=512, width=512, diff=1.5))
    m.add_hole_to_phase_field(256, 256, 30)
    m.define()
    ref = state(m)
    advance(m, 100)
    orc.fenton_run(ref, 0.1, 1.5, m.phase, 1000)
    got = state(m)
    err = np.abs(got.astype(np.float64) - ref).max()
    assert err <= 1e-3, err                       # tolerance stated for 1000 sub-steps; measured ~4e-7


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


def test_br_512_config2_vs_oracle(gpu_lib, orc):
    """br.py __main__ (br.py:348-365): 512x512, diff 0.809, cheby=True, hole (150,200,40); 100 sub-steps"""
    from fib_tf_amd.br import BeelerReuter
    m = BeelerReuter(dict(BASE, height=512, width=512, diff=0.809))
    m.add_hole_to_phase_field(150, 200, 40)
    m.define()
    ref = state(m)
    advance(m, 20)
    orc.br_run(ref, 0.1, 0.809, m.phase, m.chebyshev_table().astype(np.float32), False, 20)
    got = state(m)
    assert np.abs(got[0].astype(np.float64) - ref[0]).max() <= 2e-5 * 120.0
    assert np.abs(got[2:].astype(np.float64) - ref[2:]).max() <= 2e-5


def test_court_1024_config4_vs_oracle(gpu_lib, orc):
    """court.py protocol scaled x2: 1024x1024, two holes, fast tick + 'slow' every 10th"""
    from fib_tf_amd.court import Courtemanche
    m = Courtemanche(dict(BASE, height=1024, width=1024, diff=0.809))
    m.add_hole_to_phase_field(512, 512, 60)
    m.add_hole_to_phase_field(512, 512, 500, neg=True)
    m.define()
    ref = state(m)
    advance(m, 21, lambda i: m.fire_op('slow') if i % 10 == 0 else None)
    orc.court_run(ref, 0.1, 0.809, m.phase, True, 0, 21)
    got = state(m)
    assert np.abs(got[0].astype(np.float64) - ref[0]).max() <= 2e-5 * 150.0
    for i in range(1, 21):
        sc = max(float(np.abs(ref[i]).max()), 1e-3)
        assert np.abs(got[i].astype(np.float64) - ref[i]).max() <= 2e-5 * sc, m.VAR_NAMES[i]


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
