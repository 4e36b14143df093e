"""BeelerReuter — the modified 8-variable Beeler-Reuter ventricular model behind the reference's
API (siravan/fib_tf `br.py:31-343`).  Device side: csrc/models.hpp `BeelerReuter`.  Host side kept
here exactly where the reference keeps it: the α/β coefficient table (br.py:49-62) and, for
config['cheby'], the definition-time least-squares Chebyshev fits and their change of basis
(br.py:275-287, 303-332), done in NumPy float64 and handed to the kernel as 12x9 float32
constants."""
import os

import numpy as np

from . import _lib
from .ionic import IonicModel


SPEC_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_spec')
_spec_cache = {}


def specialised_tag(table32):
    """what names a specialised build: the table, every source file and the compiler flags"""
    import hashlib
    h = hashlib.sha1(table32.tobytes())
    for d in _lib.DEPS:
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(_lib.HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def specialised_library(table32, build=True, verbose=False):
    """libfibhip with the 12x9 Chebyshev table compiled in (csrc/br_step.inc, FIB_BR_TABLE_INC); None when it
    is neither cached nor buildable here (the caller then stays on the stock library)"""
    import subprocess
    alt = os.environ.get('FIBHIP_BR_LIBRARY')           # tuning experiments (tools/ab.sh): another build for this table
    if alt:
        return _lib.load(alt)
    tag = specialised_tag(table32)
    if tag in _spec_cache:
        return _spec_cache[tag]
    so = os.path.join(SPEC_DIR, 'libfibhip_br_%s.so' % tag)
    L = None
    try:
        if not os.path.exists(so) and build:
            os.makedirs(SPEC_DIR, exist_ok=True)
            inc = os.path.join(SPEC_DIR, 'br_table_%s.inc' % tag)
            tmp = '%s.%d.tmp' % (inc, os.getpid())
            with open(tmp, 'w') as f:
                f.write('// Chebyshev coefficients of fib_tf_amd/br.py chebyshev_table(), float32, row-major 12 x 9\n')
                f.write('static constexpr float FIB_BR_CHEB[108] = {\n    %s};\n'
                        % ',\n    '.join(', '.join(float(x).hex() + 'f' for x in row) for row in table32.reshape(12, 9)))
            os.replace(tmp, inc)
            _lib.build_specialised(['-DFIB_ONLY_BR', '-DFIB_BR_TABLE_INC="%s"' % inc], so, verbose=verbose)
        if os.path.exists(so):
            L = _lib.load(so)
    except (OSError, subprocess.CalledProcessError) as e:
        print('fib_tf_amd.br: no specialised build (%s); using the stock library' % e)
    _spec_cache[tag] = L
    return L


class BeelerReuter(IonicModel):
    MODEL_ID = _lib.BR
    VAR_NAMES = ('V', 'C', 'M', 'H', 'J', 'D', 'F', 'XI')

    def __init__(self, props):
        super().__init__(props)
        self.min_v = -90.0          # mV, br.py:42-44
        self.max_v = 30.0
        self.depol = -84.6
        for key in ('skip', 'cheby'):
            if not hasattr(self, key):
                setattr(self, key, False)
        # rows: ca_x1 cb_x1 ca_m cb_m ca_h cb_h ca_j cb_j ca_d cb_d ca_f cb_f; the d/f rows carry the
        # factor 2 that halves the calcium gate time constants (br.py:46-62)
        self.ab_coef = np.array(
            [[0.0005, 0.083, 50., 0.0, 0.0, 0.057, 1.0],
             [0.0013, -0.06, 20., 0.0, 0.0, -0.04, 1.0],
             [0.0000, 0.0, 47., -1.0, 47., -0.1, -1.0],
             [40., -0.056, 72., 0.0, 0.0, 0.0, 0.0],
             [0.126, -.25, 77., 0.0, 0.0, 0.0, 0.0],
             [1.7, 0.0, 22.5, 0.0, 0.0, -0.082, 1.0],
             [0.055, -.25, 78.0, 0.0, 0.0, -0.2, 1.0],
             [0.3, 0.0, 32., 0.0, 0.0, -0.1, 1.0],
             [2 * 0.095, -0.01, -5., 0.0, 0.0, -0.072, 1.0],
             [2 * 0.07, -0.017, 44., 0.0, 0.0, 0.05, 1.0],
             [2 * 0.012, -0.008, 28., 0.0, 0.0, 0.15, 1.0],
             [2 * 0.0065, -0.02, 30., 0.0, 0.0, -0.2, 1.0]],
            dtype=np.float32)

    def _flags(self):
        return (super()._flags() | (_lib.CHEBY if self.cheby else 0) | (_lib.SKIP if self.skip else 0)
                | (_lib.HOLD if getattr(self, '_hold', False) else 0))

    # ---- definition-time Chebyshev machinery (host, float64) ------------------------------------
    def calc_alpha_beta_np(self):
        """α and β of the six gates sampled at 1001 voltages in [min_v, max_v] (br.py:275-287)"""
        v = np.linspace(self.min_v, self.max_v, 1001)
        c = self.ab_coef
        x = np.outer(v, np.ones(c.shape[0]))
        y = ((c[:, 0] * np.exp(c[:, 1] * (x + c[:, 2])) + c[:, 3] * (x + c[:, 4])) /
             (np.exp(c[:, 5] * (x + c[:, 2])) + c[:, 6]))
        return v, y[..., ::2], y[..., 1::2]

    @staticmethod
    def leading_term_coefficients(x, y, deg=8):
        """least-squares Chebyshev fit of y(x), re-expressed in the basis S_i = 2^(i-1) x^i (the
        leading term of T_i) that the kernel evaluates (br.py:303-327)"""
        c = np.polynomial.chebyshev.Chebyshev.fit(x, y, deg).coef
        a = np.zeros([deg + 1, deg + 1], dtype=np.int64)     # a[i, j] = coefficient of x^j in T_i
        a[0, 0] = 1
        a[1, 1] = 1
        for i in range(2, deg + 1):
            a[i, 1:] += 2 * a[i - 1, :-1]
            a[i, :] -= a[i - 2, :]
        a //= np.diag(a)                                      # divide column j by the leading coefficient of T_j
        return np.matmul(np.transpose(a), c)

    def chebyshev_table(self):
        """12 x 9 float64 table in the kernel's row order
        m_inf h_inf m_tau h_tau | xi_inf j_inf d_inf f_inf | xi_tau j_tau d_tau f_tau (br.py:223-240)"""
        v, al, be = self.calc_alpha_beta_np()
        rows = []
        for kind, g in (('inf', 1), ('inf', 2), ('tau', 1), ('tau', 2),
                        ('inf', 0), ('inf', 3), ('inf', 4), ('inf', 5),
                        ('tau', 0), ('tau', 3), ('tau', 4), ('tau', 5)):
            y = al[:, g] / (al[:, g] + be[:, g]) if kind == 'inf' else 1.0 / (al[:, g] + be[:, g])
            rows.append(self.leading_term_coefficients(v, y, 8))
        return np.array(rows)

    def _configure_stepper(self, st):
        if self.cheby:
            st.set_consts(self._table32())

    def _table32(self):
        if getattr(self, '_tbl32', None) is None:
            self._tbl32 = np.ascontiguousarray(self.chebyshev_table().astype(np.float32))   # each coefficient rounded once
        return self._tbl32

    def _new_stepper(self, steps_per_tick=0, shard=True):
        # Chebyshev gates: run a build of the library that has THIS table baked in as literals (same arithmetic,
        # bit-identical results; +50 % at 512^2 because literal operands issue at full VALU rate and kernel
        # arguments do not).  Compiled once per table (hipcc, ~15 s) and cached next to the package; without a
        # compiler the stock library with the table as a kernel argument is used.  config['specialise']=False
        # forces the stock library.
        if self.cheby and getattr(self, 'specialise', True) and self._library is None \
                and not os.environ.get('FIBHIP_VARIANT'):
            # (FIBHIP_VARIANT = a tuning sweep over kernel shapes only the stock library carries)
            self._library = specialised_library(self._table32())
            if self._library is not None:
                _lib.warm(getattr(self, 'device', 0))       # the stock library's kernels first (include/fibhip.h fibhip_warm)
        return super()._new_stepper(steps_per_tick, shard)

    def define(self, s1=True, state=None):
        """initial conditions br.py:71-82 (S1: V[:,1] = 10 mV); one tick = 5 sub-steps, with
        config['skip'] the slow gates advance 5·dt on the first of them only (br.py:98-107).
        state: resume from a dict V/C/M/H/J/D/F/XI -> [H,W] array instead (`model.state` after
        `run(keep_state=True)`; the contract of court.py:87-89)"""
        IonicModel.define(self)
        if state is not None:
            init = self._resume_arrays(state)
        else:
            shape = [self.height, self.width]
            init = [np.full(shape, v, dtype=np.float32)
                    for v in (-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)]
            if s1:
                init[0][:, 1] = 10.0
        self._create(init)
        self._V = self._State['V']

    def solve(self, state, n=1):
        """ONE sub-step of (V, C, M, H, J, D, F, XI) host arrays on the GPU (br.py:125-173).
        n = number of dt the slow gates advance: 1; 5 (the first sub-step of a skip tick); 0 (its other four
        sub-steps: xi, j, d, f are carried over unchanged, br.py:98-103)."""
        if n not in (0, 1, 5):
            raise ValueError('solve: n must be 0, 1 or 5')
        keep = self.skip
        self.skip, self._hold = (n == 5), (n == 0)
        try:
            st = self._new_stepper(steps_per_tick=1, shard=False)
        finally:
            self.skip, self._hold = keep, False
        try:
            st.set_state(-1, np.stack([np.asarray(a, np.float32) for a in state]))
            st.step(1)
            return tuple(st.get_state(-1))
        finally:
            st.close()

    def pot(self):
        return self._V

    def image(self):
        """V scaled to 0..1 (br.py:337-343)"""
        v = self._V.eval()
        return (v - self.min_v) / (self.max_v - self.min_v)
