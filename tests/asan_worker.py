"""child process of tests/test_oracle_sanitizers.py: the oracle built with AddressSanitizer + UBSan (FIB_ORACLE_LIB,
LD_PRELOAD=libasan) through its unit ops, single steps, trajectories and odd sizes.  Any invalid access or undefined
operation aborts the process; the parent also checks the numbers this prints."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc

GOLD = os.path.join(ROOT, 'tests', 'golden')
orc.set_threads(2)


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def close(got, want, tol, what):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max())
    assert err <= tol, '%s: %g' % (what, err)


u = golden('unit_ops')                                                        # 37 x 53
assert np.array_equal(orc.enforce_boundary(u['X']), u['enforce_boundary'])
assert np.array_equal(orc.laplace(u['X']), u['laplace_nophase'])
assert np.array_equal(orc.laplace(u['X'], u['phi']), u['laplace_phase'])
assert np.array_equal(orc.phase_field(u['X'], u['phi']), u['phase_field'])
close(orc.rush_larsen(u['rl_g'], u['rl_inf'], u['rl_tau'], 0.1), u['rush_larsen_dt0.1'], 3e-7, 'rush_larsen')

for variant in ('phase', 'nophase'):
    f = golden('fenton_step_' + variant)
    out = orc.fenton_step(np.stack([f[k] for k in 'UVWS']), float(f['dt']), float(f['diff']), f['phase'])
    for i, k in enumerate('UVWS'):
        close(out[i], f[k + '1'], 2e-7, 'fenton step ' + k)
f = golden('br_step')
tbl = golden('br_cheby_table')['d'].astype(np.float32)
for mode, t in (('direct', None), ('cheby', tbl)):
    for n in (0, 1, 5):
        out = orc.br_step(np.stack([f[k] for k in orc.BR_VARS]), 0.1, 0.809, f['phase'], t, n)
        close(out[0], f['V1_%s_n%d' % (mode, n)], 3e-4, 'br step V %s %d' % (mode, n))
f = golden('court_step')
keys = [k for k in f.files]
slab = np.stack([f[k] for k in orc.COURT_VARS]) if all(k in keys for k in orc.COURT_VARS) else None
if slab is not None:
    out = orc.court_step(slab, 0.1, 0.809, f['phase'] if 'phase' in keys else None, True)
    assert np.isfinite(out[0]).all()

# 64 x 64 trajectories against golden
f = golden('fenton_traj64')
slab = np.stack([f['init_' + k] for k in 'UVWS']).astype(np.float32)
t0 = 0
for t in [int(x) for x in f['snap_ticks']][:3]:
    orc.fenton_run(slab, 0.1, float(f['diff']), f['phase'], (t - t0) * 10)
    t0 = t
    close(slab[0], f['U_t%d' % t], 1e-5, 'fenton traj U t%d' % t)
f = golden('court_traj64')
print('court fixture keys', len(f.files))

# odd sizes: 3 x 5 (the smallest grid the pads allow), 37 x 53, with and without a phase field
rng = np.random.default_rng(5)
for H, W in ((3, 5), (37, 53), (5, 3), (4, 64)):
    phi = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
    for p in (None, phi):
        s4 = rng.uniform(0, 1, (4, H, W)).astype(np.float32)
        orc.fenton_run(s4, 0.1, 1.5, p, 20)
        assert np.isfinite(s4).all()
        s8 = np.empty((8, H, W), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            s8[i] = v
        s8[0] += rng.uniform(0, 60, (H, W)).astype(np.float32)
        orc.br_run(s8, 0.1, 0.809, p, tbl, False, 3)
        orc.br_run(s8, 0.1, 0.809, p, None, True, 3)
        assert np.isfinite(s8).all()
        s21 = np.empty((21, H, W), np.float32)
        from fib_tf_amd.court import INITIAL
        for i, (_, v) in enumerate(INITIAL):
            s21[i] = v
        s21[0] += rng.uniform(0, 60, (H, W)).astype(np.float32)
        orc.court_run(s21, 0.1, 0.809, p, True, 0, 21)
        assert np.isfinite(s21).all()
        x = rng.uniform(-1, 1, (H, W)).astype(np.float32)
        orc.laplace(x, p)
        orc.enforce_boundary(x)
        orc.pace(x, 0, H // 2 + 1, 0, W // 2 + 1, 1.0, 0.0)
        s4 = rng.uniform(0, 1, (4, H, W)).astype(np.float32)
        orc.fenton_simple_run(s4, 0.1, 1.5, 5)
print('sanitizer worker: all checks passed')
