#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE — runs only in the build container.

It imports the reference's own model modules from /root/reference (read-only,
never copied) with `_standin/tensorflow` (a float32 NumPy stand-in for the TF-1
elementwise calls, see its docstring) ahead of them on sys.path, drives the
reference's OWN functions and stores inputs + outputs as small .npz fixtures in
this directory.  The reference does not travel to the GPU box; only the fixtures
and this script do.

How `Session.run(op)` is replayed without TensorFlow: under the stand-in the
reference's `define()` evaluates eagerly, so one call of the reference's
`define()` computes exactly one tick (Fenton: 10 x solve, `fenton.py:133-138`;
Beeler-Reuter: 5 x solve or the skip schedule, `br.py:98-107`; Courtemanche: one
solve split into fast/slow assign groups, `court.py:91-103`) and returns the
(variable, new value) pairs it would assign.  The generator applies the pairs
and feeds them back as the variables' current values for the next `define()`
(via `tensorflow.variable_override` / `define(state=...)`).  Pacing uses the
reference's `add_pace_op` (`ionic.py:125-163`) in the same way.

Usage:  python tests/golden/make_golden.py            (about 2-3 minutes)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('FIBTF_REFERENCE', '/root/reference')

sys.dont_write_bytecode = True                      # /root/reference is read-only
np.int = int                                        # br.py:319 uses the removed alias
sys.path[:0] = [os.path.join(HERE, '_standin'),
                os.path.join(HERE, '_standin', 'screen_stub'), REF]

import tensorflow as tf                             # noqa: E402  (the stand-in)
import ionic                                        # noqa: E402,F401  (reference)
import fenton                                       # noqa: E402  (reference)
import br                                           # noqa: E402  (reference)
import court                                        # noqa: E402  (reference)
import court_ultra                                  # noqa: E402  (reference)
import fenton_simple                                # noqa: E402  (reference)

T = tf.Tensor
f32 = np.float32


def cfg(h, w, diff, **kw):
    c = {'width': w, 'height': h, 'dt': 0.1, 'dt_per_plot': 10, 'diff': diff,
         'duration': 1000, 'timeline': False, 'timeline_name': 'unused.json',
         'save_graph': False, 'skip': False, 'cheby': False}
    c.update(kw)
    return c


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KiB' % (name + '.npz', os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------
# 1. unit ops of IonicModel (ionic.py:44-123)
# --------------------------------------------------------------------------
def unit_ops():
    rng = np.random.default_rng(1234)
    H, W = 37, 53
    X = rng.uniform(-1.0, 2.0, (H, W)).astype(f32)
    phi = rng.uniform(0.05, 1.0, (H, W)).astype(f32)
    m = ionic.IonicModel(cfg(H, W, 1.0))
    out = {'X': X, 'phi': phi}
    out['enforce_boundary'] = m.enforce_boundary(T(X)).a
    out['laplace_nophase'] = m.laplace(T(X)).a
    m.phase = phi
    out['laplace_phase'] = m.laplace(T(X)).a
    out['phase_field'] = m.phase_field(tf.pad(T(X), [[1, 1], [1, 1]], 'REFLECT')).a
    g = rng.uniform(0.0, 1.0, (H, W)).astype(f32)
    ginf = rng.uniform(-0.2, 1.2, (H, W)).astype(f32)
    tau = np.exp(rng.uniform(np.log(0.01), np.log(500.0), (H, W))).astype(f32)
    tau[0, :5] = [-1.2, -0.05, 1e-4, 1e4, 0.1]     # BR's Chebyshev fit yields tau<0
    out.update(rl_g=g, rl_inf=ginf, rl_tau=tau)
    for dt in (0.1, 0.5, 1.0):
        out['rush_larsen_dt%g' % dt] = m.rush_larsen(T(g), T(ginf), T(tau), dt).a
    # phase-field construction (host side, ionic.py:83-105)
    m2 = ionic.IonicModel(cfg(H, W, 1.0))
    m2.add_hole_to_phase_field(20, 15, 6)
    out['hole_a'] = np.array(m2.phase)
    m2.add_hole_to_phase_field(26, 18, 30, neg=True)
    out['hole_ab'] = np.array(m2.phase)
    save('unit_ops', **out)


# --------------------------------------------------------------------------
# 2. single-step, branch-covering
# --------------------------------------------------------------------------
def fenton_step():
    rng = np.random.default_rng(1234)
    H, W = 37, 53
    U = rng.uniform(-0.05, 1.05, (H, W)).astype(f32)
    # exact thresholds: sign()==0 -> Heaviside 0.5 (fenton.py:73-79); strict > (fenton.py:87-88)
    U[3, 3:9] = [0.23, 0.3, 0.146, 0.84, 0.8, 0.0]
    U[0, 7] = 0.23
    U[5, 0] = 0.3
    V, Wg, S = (rng.uniform(0.0, 1.0, (H, W)).astype(f32) for _ in range(3))
    for diff, hole in ((1.5, True), (0.7, False)):
        m = fenton.Fenton4v(cfg(H, W, diff))
        if hole:
            m.add_hole_to_phase_field(25, 17, 7)
        U1, V1, W1, S1 = m.solve((T(U), T(V), T(Wg), T(S)))
        dU, dV, dW, dS = m.differentiate(T(U), T(V), T(Wg), T(S))
        save('fenton_step_%s' % ('phase' if hole else 'nophase'),
             diff=diff, dt=0.1, phase=(m.phase if hole else np.zeros(0, f32)),
             U=U, V=V, W=Wg, S=S, U1=U1.a, V1=V1.a, W1=W1.a, S1=S1.a,
             dU=dU.a, dV=dV.a, dW=dW.a, dS=dS.a)


BR_NAMES = ['V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI']


def br_random_state(rng, H, W):
    st = {'V': rng.uniform(-90.0, 30.0, (H, W)).astype(f32),
          'C': np.exp(rng.uniform(np.log(1e-7), np.log(1e-5), (H, W))).astype(f32)}
    for n in BR_NAMES[2:]:
        st[n] = rng.uniform(1e-5, 0.99999, (H, W)).astype(f32)
    st['V'][2, 2:6] = [-85.0, 25.0, -89.99, 29.99]
    return st


def br_step():
    rng = np.random.default_rng(4321)
    H, W = 37, 53
    st = br_random_state(rng, H, W)
    out = {k: v for k, v in st.items()}
    for cheby in (False, True):
        m = br.BeelerReuter(cfg(H, W, 0.809, cheby=cheby))
        m.add_hole_to_phase_field(25, 17, 7)
        out['phase'] = m.phase
        for n in (1, 5, 0):
            res = m.solve(tuple(T(st[k]) for k in BR_NAMES), n)
            for k, r in zip(BR_NAMES, res):
                out['%s1_%s_n%d' % (k, 'cheby' if cheby else 'direct', n)] = r.a
    save('br_step', diff=0.809, dt=0.1, **out)


class _Term:
    """records `d[i] * Ts[i]` inside the reference's expand_chebyshev (br.py:329-331)."""
    __array_ufunc__ = None

    def __init__(self, i, sink):
        self.i, self.sink = i, sink

    def __rmul__(self, c):
        self.sink[self.i] = float(c)
        return self

    def __radd__(self, other):                      # np.float64(d[0]) + term
        self.sink[0] = float(other)
        return self

    def __iadd__(self, other):                      # (running sum) += term
        return self


def br_cheby_table():
    """the 12 x 9 coefficient table `d` of br.py:327 in the order the device needs
    (rows: m_inf,h_inf,m_tau,h_tau, xi_inf,j_inf,d_inf,f_inf, xi_tau,j_tau,d_tau,f_tau;
    br.py:223-240)"""
    m = br.BeelerReuter(cfg(8, 8, 0.809, cheby=True))
    v, al, be = m.calc_alpha_beta_np()
    rows = []
    sel = [('inf', 1), ('inf', 2), ('tau', 1), ('tau', 2),
           ('inf', 0), ('inf', 3), ('inf', 4), ('inf', 5),
           ('tau', 0), ('tau', 3), ('tau', 4), ('tau', 5)]
    for kind, g in sel:
        y = al[:, g] / (al[:, g] + be[:, g]) if kind == 'inf' else 1.0 / (al[:, g] + be[:, g])
        sink = np.zeros(9)
        Ts = [1.0] + [_Term(i, sink) for i in range(1, 9)]
        m.expand_chebyshev(Ts, v, y)
        rows.append(sink)
    d = np.array(rows)
    save('br_cheby_table', d=d, v=v, alpha=al, beta=be)


COURT_NAMES = ['V', '_Na_i_', '_m_', '_h_', '_j_', '_K_i_', '_oa_', '_oi_', '_ua_', '_ui_',
               '_xr_', '_xs_', '_Ca_i_', '_d_', '_f_', '_f_Ca_', '_Ca_rel_', '_u_', '_v_',
               '_w_', '_Ca_up_']


def court_random_state(rng, H, W):
    st = {}
    for n in COURT_NAMES:
        st[n] = rng.uniform(1e-5, 0.99999, (H, W)).astype(f32)
    st['V'] = rng.uniform(-100.0, 50.0, (H, W)).astype(f32)
    st['_Na_i_'] = rng.uniform(10.0, 13.0, (H, W)).astype(f32)
    st['_K_i_'] = rng.uniform(130.0, 145.0, (H, W)).astype(f32)
    st['_Ca_i_'] = np.exp(rng.uniform(np.log(5e-5), np.log(1e-3), (H, W))).astype(f32)
    st['_Ca_rel_'] = rng.uniform(0.5, 2.0, (H, W)).astype(f32)
    st['_Ca_up_'] = rng.uniform(1.0, 2.0, (H, W)).astype(f32)
    # every `where` singularity of calc_inter (court.py:303-410): the exact float32
    # value, both float32 neighbours and +-1e-3
    sing = [-10.0001, -10.0, 7.9, -47.13, -40.0, -14.1, 3.3328, 19.9]
    vals = []
    for s in sing:
        s32 = f32(s)
        vals += [s32, np.nextafter(s32, f32(1e9)), np.nextafter(s32, f32(-1e9)),
                 f32(s + 1e-3), f32(s - 1e-3), f32(s + 5e-4), f32(s - 5e-4)]
    vals = np.array(vals, f32)
    r, c = np.unravel_index(np.arange(len(vals)) * 7 + 60, (H, W))
    st['V'][r, c] = vals
    return st


def court_step():
    rng = np.random.default_rng(999)
    H, W = 37, 53
    st = court_random_state(rng, H, W)
    out = dict(st)
    with np.errstate(all='ignore'):
        for chronic in (True, False):
            m = court.Courtemanche(cfg(H, W, 0.809))
            m.chronic = chronic
            m.add_hole_to_phase_field(25, 17, 7)
            out['phase'] = m.phase
            res = m.solve({k: T(v) for k, v in st.items()})
            for k in COURT_NAMES:
                out['%s_1_%s' % (k, 'chronic' if chronic else 'acute')] = res[k].a
        # calc_inter through the numpy branch at the cross-check voltage of
        # generate_table.cpp:15 (python floats -> float64 arithmetic there)
        m = court.Courtemanche(cfg(H, W, 0.809))
        inter = m.calc_inter(T(st['V']), tf)
        for k, v in inter.items():
            out['inter_' + k] = v.a
    save('court_step', diff=0.809, dt=0.1, **out)


# --------------------------------------------------------------------------
# 3. trajectories through the reference's define()
# --------------------------------------------------------------------------
def apply_pairs(pairs):
    for var, new in pairs:
        tf.variable_override[var.name] = np.array(new.a)


def current(names):
    return {n: np.array(tf.variable_override[n]) for n in names}


def fenton_traj(name, H, W, diff, hole, ticks, snaps, s2=None, cube_every=None):
    tf.variable_override.clear()
    m = fenton.Fenton4v(cfg(H, W, diff))
    if hole:
        m.add_hole_to_phase_field(*hole)
    out = {'phase': m.phase if hole else np.zeros(0, f32), 'diff': diff, 'dt': 0.1,
           'hole': np.array(hole if hole else [], f32)}
    cube = []
    for i in range(ticks):
        m.define()                                  # == sess.run(_ode_op), ionic.py:203
        if i == 0:
            for var, _ in m._ode_op:
                out['init_' + var.name] = np.array(var.a)
        apply_pairs(m._ode_op)
        if s2 is not None and i == s2[0]:           # fire_op after the tick, fenton.py:182-183
            m._U = T(tf.variable_override['U'], name='U')
            m.add_pace_op('s2', s2[1], s2[2])
            apply_pairs([m._ops['s2']])
        if cube_every and i % cube_every == 0:      # fenton.py:184-185
            img = np.array(tf.variable_override['U'])
            cube.append(img * m.phase if hole else img)
        if (i + 1) in snaps:
            for k, v in current('UVWS').items():
                out['%s_t%d' % (k, i + 1)] = v
    out['snap_ticks'] = np.array(sorted(snaps))
    out['dt_per_step'] = m.dt_per_step
    if cube:
        out['cube'] = np.array(cube, f32)
        out['s2'] = np.array([s2[0], s2[2]], f32)
    save(name, **out)


def fenton_simple_traj(name, H, W, diff, steps, snaps, s2_step):
    """fenton_simple.py (= fenton_jit.py without the XLA scope): one solve per op, Laplacian by a zero-padded
    3x3 convolution (fenton_simple.py:38-49), its own S2 op `U = max(U, s2_init)` on [:H//2, :W//2] fired after
    step int(s2_time/dt) (fenton_simple.py:150-152,172,191-192)"""
    tf.variable_override.clear()
    c = {'width': W, 'height': H, 'dt': 0.1, 'dt_per_plot': 10, 'diff': diff, 'samples': steps,
         's2_time': s2_step * 0.1}
    fenton_simple.config = c                        # the constructor reads this module-level name (fenton_simple.py:71)
    m = fenton_simple.Fenton4vSimple(c)
    out = {'diff': diff, 'dt': 0.1, 's2_step': int(c['s2_time'] / c['dt'])}
    for i in range(steps):
        m.define()
        if i == 0:
            for var, _ in m._ode_op:
                out['init_' + var.name] = np.array(var.a)
        apply_pairs(m._ode_op)
        if i == out['s2_step']:
            m.define()
            apply_pairs([m._s2_op])
        if (i + 1) in snaps:
            for k, v in current('UVWS').items():
                out['%s_t%d' % (k, i + 1)] = v
    out['snap_steps'] = np.array(sorted(snaps))
    save(name, **out)


def br_traj(name, H, W, diff, hole, ticks, snaps, cheby, skip, s2=None):
    tf.variable_override.clear()
    m = br.BeelerReuter(cfg(H, W, diff, cheby=cheby, skip=skip))
    m.add_hole_to_phase_field(*hole)
    out = {'phase': m.phase, 'diff': diff, 'dt': 0.1, 'hole': np.array(hole, f32),
           'cheby': cheby, 'skip': skip}
    for i in range(ticks):
        m.define()
        if i == 0:
            for var, _ in m._ode_op:
                out['init_' + var.name] = np.array(var.a)
        apply_pairs(m._ode_op)
        if s2 is not None and i == s2[0]:
            m._V = T(tf.variable_override['V'], name='V')
            m.add_pace_op('s2', s2[1], s2[2])
            apply_pairs([m._ops['s2']])
        if (i + 1) in snaps:
            for k, v in current(BR_NAMES).items():
                out['%s_t%d' % (k, i + 1)] = v
    out['snap_ticks'] = np.array(sorted(snaps))
    out['dt_per_step'] = m.dt_per_step
    save(name, **out)


def court_traj(name, H, W, diff, holes, ticks, snaps, s2=None):
    """court.py:615-621 schedule: fast group every tick (ionic.py:203), then the
    caller fires 'slow' when i % 10 == 0 — a second evaluation of solve on the
    post-fast state."""
    m = court.Courtemanche(cfg(H, W, diff))
    for h in holes:
        m.add_hole_to_phase_field(*h)
    out = {'phase': m.phase, 'diff': diff, 'dt': 0.1}
    state = None
    trend = []
    with np.errstate(all='ignore'):
        for i in range(ticks):
            m.defined = False
            m.define(state=state)
            if i == 0:
                state = {k: np.array(v.a) for k, v in m._State.items()}
                for k, v in state.items():
                    out['init_' + k] = v
                names = list(m._State.keys())
                var2name = {id(v): k for k, v in m._State.items()}
            else:
                var2name = {id(v): k for k, v in m._State.items()}
            for var, new in m._ode_op:
                state[var2name[id(var)]] = np.array(new.a)
            if i % 10 == 0:
                m.define(state=state)
                var2name = {id(v): k for k, v in m._State.items()}
                for var, new in m._ops['slow']:
                    state[var2name[id(var)]] = np.array(new.a)
                # 'trend' probe, court.py:107-111
                trend.append([state['V'][W // 2, 20], state['_Na_i_'][W // 2, 20]])
            if s2 is not None and i == s2[0]:
                m._V = T(state['V'])
                m.add_pace_op('s2', s2[1], s2[2])
                state['V'] = np.array(m._ops['s2'][1].a)
            if (i + 1) in snaps:
                for k in names:
                    out['%s_t%d' % (k, i + 1)] = np.array(state[k])
    out['snap_ticks'] = np.array(sorted(snaps))
    out['trend'] = np.array(trend, f32)
    out['names'] = np.array(names)
    save(name, **out)


def court_ultra_traj(name, H, W, diff, holes, ticks, snaps, s2=None, ultra_slow=False):
    """court_ultra.py: every tick assigns all variables with dt (court_ultra.py:107-111,127-128);
    config['ultra_slow'] = False as in its own __main__ (court_ultra.py:543), or True for the
    22-variable model with the `_us_` gate (court_ultra.py:81-82,198-199,221-222,445-450)"""
    m = court_ultra.Courtemanche(cfg(H, W, diff, ultra_slow=ultra_slow))
    for h in holes:
        m.add_hole_to_phase_field(*h)
    out = {'phase': m.phase, 'diff': diff, 'dt': 0.1}
    state = None
    with np.errstate(all='ignore'):
        for i in range(ticks):
            m.defined = False
            m.define(state=state)
            var2name = {id(v): k for k, v in m._State.items()}
            if i == 0:
                state = {k: np.array(v.a) for k, v in m._State.items()}
                for k, v in state.items():
                    out['init_' + k] = v
                names = list(m._State.keys())
                assert len(m._ops['slow']) == 0
            for var, new in m._ode_op:
                state[var2name[id(var)]] = np.array(new.a)
            if s2 is not None and i == s2[0]:
                m._V = T(state['V'])
                m.add_pace_op('s2', s2[1], s2[2])
                state['V'] = np.array(m._ops['s2'][1].a)
            if (i + 1) in snaps:
                for k in names:
                    out['%s_t%d' % (k, i + 1)] = np.array(state[k])
    out['snap_ticks'] = np.array(sorted(snaps))
    out['names'] = np.array(names)
    if ultra_slow:
        # the two extra intermediates on a voltage sweep (court_ultra.py:445-450), via the reference's own calc_inter
        vs = np.linspace(-100, 50, 301).astype(f32)
        with np.errstate(all='ignore'):
            inter = m.calc_inter(T(vs), tf)
        out['us_sweep_V'] = vs
        out['us_sweep_us_infinity'] = np.array(inter['us_infinity'].a)
        out['us_sweep_tau_us'] = np.array(inter['tau_us'].a)
    save(name, **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'court_ultra':
        court_ultra_traj('court_ultra_traj', 48, 56, 1.5, [(28, 24, 5), (28, 24, 22, True)], 120, {1, 10, 60, 120},
                         s2=(50, 'luq', 10.0))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'fenton_simple':
        fenton_simple_traj('fenton_simple_traj', 40, 56, 1.5, 300, {1, 2, 10, 100, 200, 300}, 150)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'court_ultra_us':
        court_ultra_traj('court_ultra_us_traj', 40, 48, 1.5, [(24, 20, 5)], 120, {1, 10, 60, 120},
                         s2=(50, 'luq', 10.0), ultra_slow=True)
        return
    unit_ops()
    fenton_step()
    br_step()
    br_cheby_table()
    court_step()
    # Fenton: 64x64, hole, 20 ticks = 200 steps; plus a no-phase ragged grid
    fenton_traj('fenton_traj64', 64, 64, 1.5, (32, 32, 6), 100, {1, 2, 10, 20, 50, 100})
    fenton_traj('fenton_traj_ragged', 45, 70, 1.1, None, 20, {1, 5, 20})
    # driver-level: S2 in the left upper quadrant + image cube (fenton.py:169-185)
    fenton_traj('fenton_driver96', 96, 96, 1.5, (48, 48, 8), 60, {60}, s2=(30, 'luq', 1.0),
                cube_every=10)
    for cheby in (False, True):
        br_traj('br_traj64_%s' % ('cheby' if cheby else 'direct'), 64, 64, 0.809, (20, 30, 6),
                40, {1, 4, 20, 40}, cheby, False)
    br_traj('br_traj64_skip', 64, 64, 0.809, (20, 30, 6), 40, {1, 4, 20, 40}, False, True)
    br_traj('br_traj64_cheby_skip', 64, 64, 0.809, (20, 30, 6), 20, {1, 20}, True, True,
            s2=(10, 'luq', 10.0))
    court_traj('court_traj64', 64, 64, 0.809, [(32, 32, 6)], 300, {1, 2, 10, 11, 100, 300})
    court_traj('court_traj_ragged', 40, 56, 0.809, [(28, 20, 5), (28, 20, 24, True)], 60,
               {1, 11, 60}, s2=(30, 'luq', 10.0))
    court_ultra_traj('court_ultra_traj', 48, 56, 1.5, [(28, 24, 5), (28, 24, 22, True)], 120, {1, 10, 60, 120},
                     s2=(50, 'luq', 10.0))
    court_ultra_traj('court_ultra_us_traj', 40, 48, 1.5, [(24, 20, 5)], 120, {1, 10, 60, 120},
                     s2=(50, 'luq', 10.0), ultra_slow=True)
    fenton_simple_traj('fenton_simple_traj', 40, 56, 1.5, 300, {1, 2, 10, 100, 200, 300}, 150)


if __name__ == '__main__':
    main()
