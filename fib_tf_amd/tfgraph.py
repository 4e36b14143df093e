"""tfgraph — the handful of TensorFlow-1 calls the reference's model files use (`tf.Variable`, `tf.assign`,
`tf.group`, `tf.where`, `tf.tanh`, ... ; inventory in SURVEY.md §8f.4), recorded as a small expression graph
instead of executed.  `fib_tf_amd.traced` turns that graph into HIP source for the fused stepping kernels, so
a model written in the reference's style runs on the MI355X without hand-written device code:

    import fib_tf_amd.tfgraph as tf            # was: import tensorflow as tf
    from fib_tf_amd.traced import IonicModel   # was: from ionic import IonicModel

or, for model files that must stay byte-for-byte unchanged, `fib_tf_amd.tfgraph.install()` registers this
module as `tensorflow` (and `fib_tf_amd.traced` / `fib_tf_amd.screen` as `ionic` / `screen`) in `sys.modules`.

Semantics follow TF's: a Python or NumPy scalar meeting a tensor is converted to float32 at that moment
(products of Python numbers are formed in double by Python itself, exactly as in the reference), and every
op is one float32 operation.  Nothing here computes: `Variable.eval()` reads the array back from the device
once the owning model has been compiled."""
import contextlib
import sys

import numpy as np

float32 = np.float32

# ops whose result is a boolean mask
_MASK_OPS = frozenset(('gt', 'ge', 'lt', 'le', 'eq', 'ne', 'and', 'or', 'not'))


def _scalar(x):
    """a Python/NumPy scalar as the float32 value TF would convert it to (kept as a Python float)"""
    return float(np.float32(x))


def _is_scalar(x):
    return isinstance(x, (int, float, np.integer, np.floating, bool, np.bool_)) or \
        (isinstance(x, np.ndarray) and x.ndim == 0)


class Tensor:
    """one node of the expression graph: `op` applied to `args` (Tensors or float scalars)"""

    __array_ufunc__ = None              # NumPy scalars/arrays defer to our reflected operators
    __array_priority__ = 1000
    __hash__ = object.__hash__

    def __init__(self, op, args=(), attr=None, name=None):
        self.op, self.args, self.attr, self.name = op, tuple(args), attr, name

    @property
    def is_mask(self):
        return self.op in _MASK_OPS

    def __repr__(self):
        return '<Tensor %s%s>' % (self.op, ' ' + self.name if self.name else '')

    # ---- arithmetic ------------------------------------------------------------------------------
    @staticmethod
    def _wrap(x):
        if isinstance(x, Tensor):
            return x
        if _is_scalar(x):
            return _scalar(x)
        raise TypeError('tfgraph: only scalars and tensors can meet a tensor in an expression, got %s '
                        '(array-valued constants belong in a tf.Variable)' % type(x).__name__)

    def _bin(self, op, other, swap=False):
        o = Tensor._wrap(other)
        return Tensor(op, (o, self) if swap else (self, o))

    def __add__(self, o): return self._bin('add', o)
    def __radd__(self, o): return self._bin('add', o, True)
    def __sub__(self, o): return self._bin('sub', o)
    def __rsub__(self, o): return self._bin('sub', o, True)
    def __mul__(self, o): return self._bin('mul', o)
    def __rmul__(self, o): return self._bin('mul', o, True)
    def __truediv__(self, o): return self._bin('div', o)
    def __rtruediv__(self, o): return self._bin('div', o, True)
    def __neg__(self): return Tensor('neg', (self,))
    def __pos__(self): return self
    def __pow__(self, e): return pow(self, e)
    def __abs__(self): return Tensor('abs', (self,))

    def __gt__(self, o): return self._bin('gt', o)
    def __ge__(self, o): return self._bin('ge', o)
    def __lt__(self, o): return self._bin('lt', o)
    def __le__(self, o): return self._bin('le', o)
    # == and != keep Python's identity semantics (graph algorithms hash nodes); use tf.equal/not_equal

    def __and__(self, o): return Tensor('and', (self, o))
    def __or__(self, o): return Tensor('or', (self, o))
    def __invert__(self): return Tensor('not', (self,))

    def __bool__(self):
        raise TypeError('tfgraph: the truth value of a symbolic tensor is undefined; use tf.where')

    def __getitem__(self, idx):
        return Tensor('index', (self,), attr=idx)

    def assign(self, value, name=None):
        return assign(self, value, name=name)

    def eval(self, session=None):
        raise TypeError('tfgraph: only tf.Variable can be evaluated')


class Variable(Tensor):
    """state array (or a small host-side array such as the reference's `Trend` probe variable)"""

    def __init__(self, initial_value, name=None, dtype=None, trainable=None):
        super().__init__('var', (), name=name)
        self.init = np.array(initial_value, dtype=np.float32)
        self.shape = self.init.shape
        self._owner = None              # the traced model, once compiled
        self._slot = None               # index of the array in the device slab
        self._host = None               # value of a host-side (non-grid) variable

    def eval(self, session=None):
        if self._owner is not None:
            return self._owner._eval_variable(self)
        return self.init.copy()

    def __array__(self, dtype=None, copy=None):
        a = self.eval()
        return a if dtype is None else a.astype(dtype)


class Assign:
    def __init__(self, ref, value, name=None):
        if not isinstance(ref, Tensor) or ref.op not in ('var', 'index'):
            raise TypeError('tf.assign: the target must be a tf.Variable (or an element of one)')
        self.ref, self.value, self.name = ref, Tensor._wrap(value), name


class Group:
    """tf.group(...): the unit `sess.run` / `fire_op` executes"""

    def __init__(self, ops, name=None):
        self.assigns = []
        for o in ops:
            if isinstance(o, Group):
                self.assigns.extend(o.assigns)
            elif isinstance(o, Assign):
                self.assigns.append(o)
            elif o is None:
                continue
            else:
                raise TypeError('tf.group: expected assign ops, got %r' % (o,))
        self.name = name

    def __len__(self):
        return len(self.assigns)


# ---- the functional surface ---------------------------------------------------------------------
def assign(ref, value, name=None, **_):
    return Assign(ref, value, name)


def group(*ops, name=None, **_):
    return Group(ops, name)


def no_op(name=None):
    return Group((), name)


def global_variables_initializer():
    return Group(())


def constant(value, dtype=None, shape=None, name=None):
    if _is_scalar(value):
        return _scalar(value)
    if dtype is None or dtype == 1 or dtype is float32:       # (1 = TF's DT_FLOAT enum, fenton_simple.py:36)
        dtype = np.float32
    return np.array(value, dtype=dtype)


_NP_UNARY = {'sign': np.sign, 'tanh': np.tanh, 'exp': np.exp, 'expm1': np.expm1, 'log': np.log, 'sqrt': np.sqrt,
             'square': np.square, 'abs': np.abs, 'neg': np.negative,
             'reciprocal': lambda v: np.float32(1) / v}


def _un(op):
    def f(x, name=None):
        if isinstance(x, Tensor):
            return Tensor(op, (x,))
        if _is_scalar(x) and op in _NP_UNARY:
            # TF turns the scalar into a float32 constant and applies the op to it (court.py:310: tf.square(0.0337))
            with np.errstate(all='ignore'):
                return float(_NP_UNARY[op](np.float32(x)))
        raise TypeError('tf.%s: expected a tensor' % op)
    f.__name__ = op
    return f


sign = _un('sign')
tanh = _un('tanh')
exp = _un('exp')
expm1 = _un('expm1')
log = _un('log')
sqrt = _un('sqrt')
square = _un('square')
abs = _un('abs')                                                    # noqa: A001  (TF's own name)
reciprocal = _un('reciprocal')
negative = _un('neg')
logical_not = _un('not')


def sigmoid(x, name=None):
    return reciprocal(1.0 + exp(-x))


def pow(x, y, name=None):                                           # noqa: A001
    if not isinstance(x, Tensor) or not _is_scalar(y):
        raise TypeError('tf.pow: tensor ** scalar only')
    return Tensor('pow', (x, _scalar(y)))


def _bin(op):
    def f(x, y, name=None):
        x, y = Tensor._wrap(x), Tensor._wrap(y)
        if not isinstance(x, Tensor) and not isinstance(y, Tensor):
            raise TypeError('tf.%s: at least one operand must be a tensor' % op)
        return Tensor(op, (x, y))
    f.__name__ = op
    return f


add = _bin('add')
subtract = _bin('sub')
multiply = _bin('mul')
divide = _bin('div')
truediv = divide
maximum = _bin('maximum')
minimum = _bin('minimum')
greater = _bin('gt')
greater_equal = _bin('ge')
less = _bin('lt')
less_equal = _bin('le')
equal = _bin('eq')
not_equal = _bin('ne')
logical_and = _bin('and')
logical_or = _bin('or')


def where(condition, x=None, y=None, name=None):
    if not isinstance(condition, Tensor) or not condition.is_mask:
        raise TypeError('tf.where: the condition must be a comparison of tensors')
    if x is None or y is None:
        raise NotImplementedError('tf.where(cond) without branches is not a pointwise op')
    return Tensor('where', (condition, Tensor._wrap(x), Tensor._wrap(y)))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return Tensor('clip', (t, Tensor._wrap(clip_value_min), Tensor._wrap(clip_value_max)))


def zeros_like(t, name=None):
    return t * 0.0


def ones_like(t, name=None):
    return t * 0.0 + 1.0


def identity(t, name=None):
    return t


def pad(tensor, paddings=None, mode='CONSTANT', name=None, **k):
    """the one pad a model file needs on its own: the no-flux boundary written out as the reference's
    `enforce_boundary` writes it — `tf.pad(X[1:-1, 1:-1], [[1, 1], [1, 1]], 'SYMMETRIC')` (ionic.py:107-113,
    fenton_simple.py:51-56).  It becomes the same graph node as IonicModel.enforce_boundary(X)."""
    interior = (slice(1, -1, None), slice(1, -1, None))
    ok = (isinstance(tensor, Tensor) and tensor.op == 'index' and tuple(tensor.attr) == interior
          and paddings is not None and np.array_equal(np.asarray(paddings), [[1, 1], [1, 1]])
          and str(mode).upper() == 'SYMMETRIC')
    if not ok:
        raise NotImplementedError('tfgraph: tf.pad is only understood as the no-flux boundary '
                                  'tf.pad(X[1:-1, 1:-1], [[1, 1], [1, 1]], \'SYMMETRIC\'); other stencils are available '
                                  'through IonicModel.enforce_boundary / IonicModel.laplace or a 3x3 tf.nn.depthwise_conv2d')
    return Tensor('bnd', (tensor.args[0],))


class _View4:
    """a [H, W] tensor seen as [1, H, W, 1] (tf.expand_dims twice) — only as the argument and the result of the 3x3
    convolution below"""

    def __init__(self, base, axes):
        self.base, self.axes = base, tuple(axes)

    def __getitem__(self, idx):
        if len(self.axes) == 2 and tuple(idx) == (0, slice(None), slice(None), 0):
            return self.base
        raise NotImplementedError('tfgraph: a [1, H, W, 1] view can only be sliced back with y[0, :, :, 0]')


def expand_dims(x, axis=None, name=None, dim=None):
    axis = dim if axis is None else axis
    if isinstance(x, Tensor) and axis == 0:
        return _View4(x, (0,))
    if isinstance(x, _View4) and x.axes == (0,) and axis in (-1, 3):
        return _View4(x.base, (0, -1))
    raise NotImplementedError('tfgraph: tf.expand_dims is only understood as expand_dims(expand_dims(x, 0), -1), the '
                              '[1, H, W, 1] view a 2-D convolution wants')


class nn:
    """tf.nn: the 3x3 depthwise convolution the stand-alone model scripts use as their Laplacian"""

    @staticmethod
    def depthwise_conv2d(input, filter, strides, padding, name=None, **k):       # noqa: A002  (TF's own names)
        kern = np.asarray(filter, dtype=np.float32)
        if not (isinstance(input, _View4) and input.axes == (0, -1) and kern.shape == (3, 3, 1, 1)
                and list(strides) == [1, 1, 1, 1] and str(padding).upper() == 'SAME'):
            raise NotImplementedError('tfgraph: tf.nn.depthwise_conv2d is only understood as a 3x3 single-channel '
                                      'convolution of a [1, H, W, 1] view with stride 1 and padding \'SAME\'')
        return _View4(Tensor('conv3', (input.base,), attr=tuple(float(v) for v in kern.reshape(9))), (0, -1))


@contextlib.contextmanager
def device(name):
    yield


@contextlib.contextmanager
def name_scope(name, *a, **k):
    yield name


variable_scope = name_scope


class Session:
    """`with tf.Session() as sess: sess.run(op)`: ops run on the model that owns their variables"""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, *a, **k):
        if isinstance(fetches, (list, tuple)):
            return [self.run(f) for f in fetches]
        if isinstance(fetches, Variable):
            return fetches.eval()
        if isinstance(fetches, (Group, Assign)):
            g = fetches if isinstance(fetches, Group) else Group((fetches,))
            if not g.assigns:
                return None
            owner = _owner_of(g)
            if owner is None:
                raise RuntimeError('tfgraph: this op does not belong to a defined IonicModel')
            owner._run_group(g)
            return None
        raise TypeError('Session.run: cannot run %r' % (fetches,))

    def close(self):
        pass


def _owner_of(g):
    for a in g.assigns:
        ref = a.ref.args[0] if a.ref.op == 'index' else a.ref
        if getattr(ref, '_owner', None) is not None:
            return ref._owner
    return None


def install(patch_numpy=True):
    """make `import tensorflow`, `import ionic` and `import screen` in UNCHANGED reference-style model files
    resolve to this package.  patch_numpy: restore the `np.int` alias (removed in NumPy 1.24) that the
    reference's br.py / court.py still use."""
    from . import screen, traced
    me = sys.modules[__name__]
    sys.modules['tensorflow'] = me
    sys.modules['ionic'] = traced
    sys.modules['screen'] = screen
    if patch_numpy and not hasattr(np, 'int'):
        np.int = int
    return me
