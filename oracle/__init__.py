"""ctypes loader for the parity oracle (oracle/fib_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (fib_tf_amd/) never does and has no CPU fallback.

`build()` compiles libfib_oracle.so with gcc (and oracle/_ref when /root/reference is
present); `lib()` loads it.  The thin wrappers below take/return NumPy float32 SoA
slabs `[nvar, H, W]`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, 'libfib_oracle.so')
SO_ASAN = os.path.join(HERE, 'libfib_oracle_asan.so')      # `make asan`: tests/test_oracle_sanitizers.py
_lib = None

FENTON_VARS = ['U', 'V', 'W', 'S']
BR_VARS = ['V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI']
COURT_VARS = ['V', '_Na_i_', '_m_', '_h_', '_j_', '_K_i_', '_oa_', '_oi_', '_ua_', '_ui_', '_xr_',
              '_xs_', '_Ca_i_', '_d_', '_f_', '_f_Ca_', '_Ca_rel_', '_u_', '_v_', '_w_', '_Ca_up_']
COURT_FAST = [0, 1, 2, 3]           # court.py:42


def build(force=False):
    src = os.path.join(HERE, 'fib_oracle.c')
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, 'libfib_oracle.so'], stdout=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get('FIBTF_REFERENCE', '/root/reference')):
        subprocess.check_call(['make', '-C', HERE, 'ref'], stdout=subprocess.DEVNULL)


_fp = C.POINTER(C.c_float)


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get('FIB_ORACLE_LIB')           # another build of the same file (the sanitizer build)
        if not alt and not os.path.exists(SO):
            build()
        _lib = C.CDLL(alt or SO)
        i, d, l, f = C.c_int, C.c_double, C.c_long, C.c_float
        sig = {
            'orc_enforce_boundary': [i, i, _fp, _fp],
            'orc_laplace': [i, i, _fp, _fp, _fp],
            'orc_phase_field': [i, i, _fp, _fp, _fp],
            'orc_rush_larsen': [l, _fp, _fp, _fp, d, _fp],
            'orc_pace': [i, i, _fp, i, i, i, i, f, f],
            'orc_fenton_diff': [l] + [_fp] * 8,
            'orc_fenton_step': [i, i, d, d, _fp, _fp, _fp, _fp],
            'orc_br_step': [i, i, d, d, _fp, _fp, i, _fp, _fp, _fp],
            'orc_court_calc_inter': [f, _fp],
            'orc_court_step': [i, i, d, d, _fp, i, _fp, _fp, _fp],
            'orc_fenton_run': [i, i, d, d, _fp, _fp, _fp, i],
            'orc_br_run': [i, i, d, d, _fp, _fp, i, _fp, _fp, i],
            'orc_court_run': [i, i, d, d, _fp, i, _fp, _fp, i, i, i],
            'orc_court_ultra_run': [i, i, d, d, _fp, i, _fp, _fp, i],
            'orc_court_ultra_us_run': [i, i, d, d, _fp, i, _fp, _fp, i],
            'orc_court_us_inter': [f, _fp, _fp],
            'orc_laplace_zeropad': [i, i, _fp, _fp],
            'orc_fenton_simple_run': [i, i, d, d, _fp, _fp, i],
        }
        for name, args in sig.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, None
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_set_exp_ulps.argtypes = [C.c_int]
        _lib.orc_set_threads(host_cores())
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _phi(phi, H, W):
    if phi is None or np.size(phi) == 0:
        return None
    phi = _f32(phi)
    assert phi.shape == (H, W)
    return phi


def enforce_boundary(X):
    X = _f32(X); out = np.empty_like(X)
    lib().orc_enforce_boundary(X.shape[0], X.shape[1], _p(X), _p(out))
    return out


def laplace(X0, phi=None):
    X0 = _f32(X0); out = np.empty_like(X0); phi = _phi(phi, *X0.shape)
    lib().orc_laplace(X0.shape[0], X0.shape[1], _p(X0), _p(phi), _p(out))
    return out


def phase_field(X0, phi):
    X0 = _f32(X0); out = np.empty_like(X0); phi = _phi(phi, *X0.shape)
    lib().orc_phase_field(X0.shape[0], X0.shape[1], _p(X0), _p(phi), _p(out))
    return out


def rush_larsen(g, ginf, tau, dt):
    g, ginf, tau = _f32(g), _f32(ginf), _f32(tau); out = np.empty_like(g)
    lib().orc_rush_larsen(g.size, _p(g), _p(ginf), _p(tau), float(dt), _p(out))
    return out


def pace(pot, r0, r1, c0, c1, v, min_v):
    pot = _f32(pot).copy()
    lib().orc_pace(pot.shape[0], pot.shape[1], _p(pot), r0, r1, c0, c1, v, min_v)
    return pot


def fenton_diff(U, V, W, S):
    a = [_f32(x) for x in (U, V, W, S)]
    o = [np.empty_like(a[0]) for _ in range(4)]
    lib().orc_fenton_diff(a[0].size, *[_p(x) for x in a + o])
    return o




def fenton_step(slab, dt, diff, phi=None):
    slab = _f32(slab); _, H, W = slab.shape
    out = np.empty_like(slab); scr = np.empty(2 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_fenton_step(H, W, dt, diff, _p(phi), _p(slab), _p(out), _p(scr))
    return out


def br_step(slab, dt, diff, phi=None, cheb=None, nslow=1):
    slab = _f32(slab); _, H, W = slab.shape
    out = np.empty_like(slab); scr = np.empty(2 * H * W, np.float32); phi = _phi(phi, H, W)
    cheb = None if cheb is None else _f32(cheb)
    lib().orc_br_step(H, W, dt, diff, _p(phi), _p(cheb), nslow, _p(slab), _p(out), _p(scr))
    return out


def court_calc_inter(V):
    out = np.empty(30, np.float32)
    lib().orc_court_calc_inter(float(V), _p(out))
    return out


def court_step(slab, dt, diff, phi=None, chronic=True):
    slab = _f32(slab); _, H, W = slab.shape
    out = np.empty_like(slab); scr = np.empty(2 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_court_step(H, W, dt, diff, _p(phi), int(chronic), _p(slab), _p(out), _p(scr))
    return out


def fenton_run(slab, dt, diff, phi, nsteps):
    """in place; nsteps sub-steps (10 per tick, fenton.py:135-138)"""
    assert slab.dtype == np.float32 and slab.flags.c_contiguous
    _, H, W = slab.shape
    tmp = np.empty(6 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_fenton_run(H, W, dt, diff, _p(phi), _p(slab), _p(tmp), nsteps)
    return slab


def br_run(slab, dt, diff, phi, cheb, skip, nticks):
    assert slab.dtype == np.float32 and slab.flags.c_contiguous
    _, H, W = slab.shape
    tmp = np.empty(10 * H * W, np.float32); phi = _phi(phi, H, W)
    cheb = None if cheb is None else _f32(cheb)
    lib().orc_br_run(H, W, dt, diff, _p(phi), _p(cheb), int(skip), _p(slab), _p(tmp), nticks)
    return slab


def court_run(slab, dt, diff, phi, chronic, tick0, nticks, slow_every=10):
    assert slab.dtype == np.float32 and slab.flags.c_contiguous
    _, H, W = slab.shape
    tmp = np.empty(23 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_court_run(H, W, dt, diff, _p(phi), int(chronic), _p(slab), _p(tmp), tick0, nticks, slow_every)
    return slab


def court_ultra_run(slab, dt, diff, phi, chronic, nticks):
    """court_ultra.py schedule: every tick assigns all 21 variables with dt"""
    assert slab.dtype == np.float32 and slab.flags.c_contiguous
    _, H, W = slab.shape
    tmp = np.empty(23 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_court_ultra_run(H, W, dt, diff, _p(phi), int(chronic), _p(slab), _p(tmp), nticks)
    return slab


def court_ultra_us_run(slab, dt, diff, phi, chronic, nticks):
    """court_ultra.py with ultra_slow=True: 22 arrays (`_us_` last), single rate"""
    assert slab.dtype == np.float32 and slab.flags.c_contiguous and slab.shape[0] == 22
    _, H, W = slab.shape
    tmp = np.empty(24 * H * W, np.float32); phi = _phi(phi, H, W)
    lib().orc_court_ultra_us_run(H, W, dt, diff, _p(phi), int(chronic), _p(slab), _p(tmp), nticks)
    return slab


def court_us_inter(V):
    """(us_infinity, tau_us) of court_ultra.py:445-450 for an array of voltages"""
    V = np.ascontiguousarray(V, np.float32)
    a, b = np.empty_like(V), np.empty_like(V)
    x, y = C.c_float(), C.c_float()
    for k, v in enumerate(V.ravel()):
        lib().orc_court_us_inter(C.c_float(v), C.byref(x), C.byref(y))
        a.ravel()[k], b.ravel()[k] = x.value, y.value
    return a, b


def fenton_simple_run(slab, dt, diff, nsteps):
    """fenton_simple.py's step (zero-padded convolution Laplacian), nsteps times, in place"""
    assert slab.dtype == np.float32 and slab.flags.c_contiguous and slab.shape[0] == 4
    _, H, W = slab.shape
    tmp = np.empty(6 * H * W, np.float32)
    lib().orc_fenton_simple_run(H, W, dt, diff, _p(slab), _p(tmp), nsteps)
    return slab


def usable_cores():
    """every core this process may use: affinity mask and cgroup CPU quota, no further cap"""
    n = len(os.sched_getaffinity(0))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()
            if q != 'max':
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def host_cores():
    """default thread count: usable_cores() capped at the GPU box's stated CPU share (16 per GPU) —
    OpenMP's default of one thread per visible core oversubscribes there"""
    n = usable_cores()
    cap = int(os.environ.get('FIBTF_ORACLE_THREADS', '16'))
    return max(1, min(n, cap))


def set_threads(n):
    lib().orc_set_threads(int(n))


def set_exp_ulps(k):
    """test knob: every expf result of the oracle moved by k float32 neighbours (0 = plain libm)"""
    lib().orc_set_exp_ulps(int(k))


def num_threads():
    return lib().orc_num_threads()
