"""float32 NumPy stand-in for the handful of TensorFlow-1 elementwise calls the
reference's model code makes.  TEST INFRASTRUCTURE ONLY.

Purpose: `tests/golden/make_golden.py` puts this directory ahead of
/root/reference on sys.path so that the reference's *own* `solve()`,
`differentiate()`, `laplace()`, `phase_field()`, `enforce_boundary()`,
`rush_larsen()`, `calc_inter()` and `expand_chebyshev()` execute unmodified and
produce the golden vectors under tests/golden/.  TensorFlow is not installed in
the build container, so this is the closest executable form of the reference.

What it pins: the reference's operation order, constants, branch structure and
float32 rounding after every single op (TF converts every Python/NumPy scalar
that meets a tensor to float32 and every kernel rounds its result to float32).
What it does NOT pin: the last-ulp behaviour of TF/Eigen's tanh/exp/expm1/log/pow
kernels (NumPy's are used instead).

Nothing here is shipped or imported by the product; nothing here is copied from
the reference.
"""
import contextlib
import numpy as np

_f32 = np.float32


def _raw(x):
    """operand -> float32 ndarray (TF's conversion rule for mixed operands)."""
    if isinstance(x, Tensor):
        return x.a
    return np.asarray(x, dtype=_f32)


class Tensor:
    """An eagerly evaluated float32 'tensor'.  Every op rounds to float32."""
    __array_priority__ = 1000
    __array_ufunc__ = None          # make NumPy scalars defer to our reflected ops

    def __init__(self, a, name=None):
        self.a = np.asarray(a, dtype=_f32)
        self.name = name

    def assign(self, value):
        """graph-mode `var.assign(x)` -> the (variable, new value) pair; the
        golden generator applies it itself (it plays the role of Session.run)."""
        return (self, value)

    # -- structure ---------------------------------------------------------
    @property
    def shape(self):
        return self.a.shape

    def __getitem__(self, idx):
        return Tensor(self.a[idx])

    def eval(self):
        return np.array(self.a, dtype=_f32)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    # -- arithmetic (each result rounded to float32) -----------------------
    def __add__(self, o): return Tensor(self.a + _raw(o))
    def __radd__(self, o): return Tensor(_raw(o) + self.a)
    def __sub__(self, o): return Tensor(self.a - _raw(o))
    def __rsub__(self, o): return Tensor(_raw(o) - self.a)
    def __mul__(self, o): return Tensor(self.a * _raw(o))
    def __rmul__(self, o): return Tensor(_raw(o) * self.a)
    def __truediv__(self, o): return Tensor(self.a / _raw(o))
    def __rtruediv__(self, o): return Tensor(_raw(o) / self.a)
    def __neg__(self): return Tensor(-self.a)

    # -- comparisons give plain bool arrays --------------------------------
    def __gt__(self, o): return self.a > _raw(o)
    def __lt__(self, o): return self.a < _raw(o)
    def __ge__(self, o): return self.a >= _raw(o)
    def __le__(self, o): return self.a <= _raw(o)


def _un(fn):
    def op(x, name=None):
        return Tensor(fn(_raw(x)))
    return op


sign = _un(np.sign)
tanh = _un(np.tanh)
exp = _un(np.exp)
expm1 = _un(np.expm1)
log = _un(np.log)
sqrt = _un(np.sqrt)
square = _un(np.square)
abs = _un(np.abs)                                   # noqa: A001 (mirrors tf.abs)
reciprocal = _un(lambda a: _f32(1.0) / a)


def pow(x, y, name=None):                           # noqa: A001 (mirrors tf.pow)
    return Tensor(np.power(_raw(x), _raw(y)))


def maximum(x, y, name=None):
    return Tensor(np.maximum(_raw(x), _raw(y)))


def minimum(x, y, name=None):
    return Tensor(np.minimum(_raw(x), _raw(y)))


def where(cond, x, y, name=None):
    c = cond.a if isinstance(cond, Tensor) else np.asarray(cond)
    return Tensor(np.where(c, _raw(x), _raw(y)))


def clip_by_value(x, lo, hi, name=None):
    return Tensor(np.minimum(np.maximum(_raw(x), _f32(lo)), _f32(hi)))


def constant(v, dtype=None, name=None):
    a = np.asarray(v)
    if dtype in (1, _f32, 'float32'):               # fenton_simple.py:35 passes the enum value of tf.float32
        return Tensor(a.astype(_f32))
    return a                                        # paddings


def expand_dims(x, axis, name=None):
    return Tensor(np.expand_dims(_raw(x), axis))


class _NN:
    """tf.nn.depthwise_conv2d for the one use the reference makes of it (fenton_simple.py:38-49): a [1,H,W,1]
    image, a [3,3,1,1] kernel, stride 1, padding='SAME' = zero padding.  The accumulation order of TF's kernel is
    not specified; taps are accumulated in the kernel's row-major order, one float32 rounding per multiply and
    per add."""

    @staticmethod
    def depthwise_conv2d(x, k, strides, padding, name=None):
        x, k = _raw(x), _raw(k)
        assert x.ndim == 4 and x.shape[0] == 1 and x.shape[3] == 1 and k.shape == (3, 3, 1, 1)
        assert list(strides) == [1, 1, 1, 1] and padding == 'SAME'
        img = np.pad(x[0, :, :, 0], 1, mode='constant')
        H, W = x.shape[1], x.shape[2]
        acc = None
        for a in range(3):
            for b in range(3):
                term = (_f32(k[a, b, 0, 0]) * img[a:a + H, b:b + W]).astype(_f32)
                acc = term if acc is None else (acc + term).astype(_f32)
        return Tensor(acc[None, :, :, None])


nn = _NN()


def pad(x, paddings, mode='CONSTANT', name=None):
    p = np.asarray(paddings)
    widths = [(int(p[i, 0]), int(p[i, 1])) for i in range(p.shape[0])]
    m = {'REFLECT': 'reflect', 'SYMMETRIC': 'symmetric', 'CONSTANT': 'constant'}[mode]
    return Tensor(np.pad(_raw(x), widths, mode=m))


# name -> ndarray: the golden generator stores the current value of each named
# variable here, so that calling the reference's define() again continues from
# the present state instead of the initial conditions (define() is the only
# place the reference encodes its per-tick unrolling schedule).
variable_override = {}


def Variable(v, name=None, dtype=None):
    if name is not None and name in variable_override:
        v = variable_override[name]
    return Tensor(np.array(v, dtype=_f32), name=name)


def assign(var, value, name=None):
    return (var, value)


def group(*ops, name=None):
    return tuple(ops)


@contextlib.contextmanager
def device(name):
    yield


@contextlib.contextmanager
def name_scope(name):
    yield name
