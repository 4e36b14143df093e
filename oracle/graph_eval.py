"""graph_eval — TEST INFRASTRUCTURE: an op-by-op float32 NumPy interpreter of the expression graphs that
fib_tf_amd.traced records from a reference-style model file.  It is the parity oracle for traced models: the
HIP code generated from a graph must reproduce what this interpreter computes from the same graph.

Semantics = what TensorFlow 1.x does for the reference's graphs on CPU (and what tests/golden/_standin does
when it replays the reference's own model files): every node is one float32 NumPy operation, scalars are
float32, `enforce_boundary` / `laplace` / `phase_field` are the pad-and-slice forms of ionic.py:44-113.
Pinned by tests/test_traced_cpu.py: the reference's unchanged fenton.py / br.py / court.py, traced and
interpreted here, must reproduce the committed golden trajectories.

Only tests/ may import this module; the product never does."""
import numpy as np

f32 = np.float32


def enforce_boundary(X):
    """ionic.py:107-113: interior re-padded SYMMETRIC"""
    return np.pad(X[1:-1, 1:-1], 1, mode='symmetric')


def laplace(X0, phase=None):
    """ionic.py:44-60 (+ phase_field ionic.py:70-81), evaluation order as written there"""
    X = np.pad(X0, 1, mode='reflect')
    L = (X[:-2, 1:-1] + X[2:, 1:-1] + X[1:-1, :-2] + X[1:-1, 2:] +
         f32(0.5) * (X[:-2, :-2] + X[2:, :-2] + X[:-2, 2:] + X[2:, 2:]) - f32(6) * X[1:-1, 1:-1])
    if phase is not None:
        P = np.pad(phase.astype(f32), 1, mode='reflect')
        L = L + ((X[2:, 1:-1] - X[:-2, 1:-1]) * (P[2:, 1:-1] - P[:-2, 1:-1]) +
                 (X[1:-1, 2:] - X[1:-1, :-2]) * (P[1:-1, 2:] - P[1:-1, :-2])) / (f32(4) * P[1:-1, 1:-1])
    return L.astype(f32)


def conv3_same(X0, k9):
    """tf.nn.depthwise_conv2d of one channel with a 3x3 kernel, stride 1, padding 'SAME' (zeros outside the array;
    fenton_simple.py:38-49).  TensorFlow does not specify its accumulation order: this is row-major over the kernel,
    one float32 rounding per product and per sum — the order tests/golden/_standin and the device kernel use."""
    X = np.pad(np.asarray(X0, f32), 1, mode='constant')
    H, W = X0.shape
    acc = None
    for i in range(3):
        for j in range(3):
            term = (f32(k9[3 * i + j]) * X[i:i + H, j:j + W]).astype(f32)
            acc = term if acc is None else (acc + term).astype(f32)
    return acc


def _c(x):
    return x if isinstance(x, np.ndarray) else f32(x)


_BIN = {'add': np.add, 'sub': np.subtract, 'mul': np.multiply, 'div': np.divide, 'maximum': np.maximum,
        'minimum': np.minimum, 'gt': np.greater, 'ge': np.greater_equal, 'lt': np.less, 'le': np.less_equal,
        'eq': np.equal, 'ne': np.not_equal, 'and': np.logical_and, 'or': np.logical_or}
_UN = {'neg': np.negative, 'sign': np.sign, 'tanh': np.tanh, 'exp': np.exp, 'expm1': np.expm1, 'log': np.log,
       'sqrt': np.sqrt, 'square': np.square, 'abs': np.abs, 'not': np.logical_not,
       'reciprocal': lambda x: f32(1) / x}


class Interpreter:
    def __init__(self, compiled, phase=None):
        """compiled = model._analyze() of a fib_tf_amd.traced.IonicModel"""
        self.c = compiled
        self.phase = phase
        self.remap = compiled['remap']

    def _eval(self, node, lv, state, memo):
        if not hasattr(node, 'op'):
            return f32(node)
        k = id(node)
        if k in memo:
            return memo[k]
        op, a = node.op, node.args
        ev = lambda x: self._eval(x, lv, state, memo)     # noqa: E731
        if op == 'param':
            r = state[self.remap[lv.bind[node.attr]]]
        elif op == 'var':
            r = state[node._slot]
        elif op == 'bnd':
            r = enforce_boundary(ev(a[0]))
        elif op == 'lap':
            r = laplace(ev(a[0]), self.phase)
        elif op == 'conv3':
            r = conv3_same(ev(a[0]), node.attr)
        elif op in _BIN:
            r = _BIN[op](_c(ev(a[0])), _c(ev(a[1])))
        elif op in _UN:
            r = _UN[op](ev(a[0]))
        elif op == 'pow':
            r = np.power(ev(a[0]), f32(a[1]))
        elif op == 'where':
            x, y = _c(ev(a[1])), _c(ev(a[2]))
            r = np.where(ev(a[0]), x, y)
        elif op == 'clip':
            r = np.clip(ev(a[0]), _c(ev(a[1])), _c(ev(a[2])))
        else:
            raise NotImplementedError(op)
        if isinstance(r, np.ndarray) and r.dtype != np.bool_:
            r = r.astype(f32, copy=False)
        memo[k] = r
        return r

    def run_mode(self, state, mode=0):
        """one execution of assign group `mode` (0 = the tick op) on state [nvar, H, W] (slab order); returns
        the new state"""
        _, prog = self.c['programs'][mode]
        state = [np.array(s, dtype=f32) for s in state]
        with np.errstate(all='ignore'):
            for lv in prog.levels:
                memo = {}
                new = {self.remap[p]: self._eval(e, lv, state, memo) for p, e in lv.outs.items()}
                for slot, val in new.items():
                    state[slot] = np.broadcast_to(val, state[slot].shape).astype(f32)
        return np.stack(state)

    def tick(self, state, n=1):
        for _ in range(n):
            state = self.run_mode(state, 0)
        return state
