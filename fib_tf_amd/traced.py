"""traced — run a USER-WRITTEN model in the reference's style on the fused HIP kernels.

The reference's model files (`fenton.py`, `br.py`, `court.py`) are ordinary Python: a subclass of
`IonicModel` whose `solve(state)` builds one explicit sub-step out of `tf.*` calls plus
`self.enforce_boundary`, `self.laplace` and `self.rush_larsen`, and whose `define()` creates `tf.Variable`s,
chains `solve` a few times and groups the assignments into `self._ode_op` (`fenton.py:95-147`).  With
`fib_tf_amd.tfgraph` standing in for `tensorflow`, this module's `IonicModel`

  1. records that graph (each distinct `solve(...)` signature is traced once; the chain in `define()` is
     recognised as "sub-step k of the tick"),
  2. emits the per-cell arithmetic as a `struct Custom` in a generated header — one C++ statement per graph node,
     in the reference's evaluation order, written over the R-wide value type of csrc/models.hpp,
  3. compiles csrc/fibhip.hip with that header into `_traced/libfibhip_<hash>.so` (hipcc, a few seconds, cached),
  4. and drives it through the same C ABI and the same `run()/fire_op()/image()` host code as the stock models.

The whole tick — every chained `solve`, the 9-point Laplacian, the phase field — is then ONE temporally
blocked kernel launch, exactly like the hand-written Fenton kernel.

Supported graph: pointwise float32 arithmetic (tfgraph's op list), one diffusing variable that enters the
stencil as `self.laplace(self.enforce_boundary(X))`, any number of further assign groups that do not use the
Laplacian (`_ops['slow']`-style), element probes into small host variables (`Trend`).  Everything else raises
NotImplementedError at define/compile time — there is no silent fallback and no CPU path."""
import hashlib
import json
import math
import os
import re

import numpy as np

from . import _lib
from . import tfgraph as tf
from .ionic import IonicModel as _HostModel
from .tfgraph import Assign, Group, Tensor, Variable

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE = os.path.join(HERE, '_traced')
MAX_VARS = 32
MAX_MODES = 8


class TraceError(NotImplementedError):
    pass


# ---------------------------------------------------------------------------------------------------
# tracing solve()
# ---------------------------------------------------------------------------------------------------
def _flatten(x, leaves):
    if isinstance(x, Tensor):
        leaves.append(x)
        return ('t',)
    if isinstance(x, dict):
        return ('d', tuple((k, _flatten(v, leaves)) for k, v in x.items()))
    if isinstance(x, (list, tuple)):
        return ('l' if isinstance(x, list) else 'u', tuple(_flatten(v, leaves) for v in x))
    return ('c', x)


def _unflatten(struct, it):
    kind = struct[0]
    if kind == 't':
        return next(it)
    if kind == 'd':
        return {k: _unflatten(s, it) for k, s in struct[1]}
    if kind in ('l', 'u'):
        vals = [_unflatten(s, it) for s in struct[1]]
        return vals if kind == 'l' else tuple(vals)
    return struct[1]


def _shape_of(struct):
    """the structure without constant payloads (cache key)"""
    kind = struct[0]
    if kind == 'd':
        return ('d', tuple((k, _shape_of(s)) for k, s in struct[1]))
    if kind in ('l', 'u'):
        return (kind, tuple(_shape_of(s) for s in struct[1]))
    if kind == 'c':
        return ('c', repr(struct[1]))
    return struct


class StepFn:
    """one traced signature of solve(): outputs as expressions of `params`"""

    def __init__(self, params, outputs, ostruct, key):
        self.params, self.outputs, self.ostruct, self.key = params, outputs, ostruct, key


class Call:
    """one application of a StepFn to concrete input tensors"""

    def __init__(self, fn, inputs):
        self.fn, self.inputs = fn, inputs


class _SolveTracer:
    """replaces `model.solve` while define() runs: every call returns opaque `apply` tensors, so the chain
    `states.append(self.solve(states[-1]))` becomes a list of Calls instead of one deep graph"""

    def __init__(self, model, orig):
        self.model, self.orig, self.cache = model, orig, {}

    def __call__(self, state, *args, **kw):
        leaves = []
        struct = _flatten(state, leaves)
        if not leaves:
            return self.orig(state, *args, **kw)
        try:
            key = (_shape_of(struct), args, tuple(sorted(kw.items())))
            hash(key)
        except TypeError:
            raise TraceError('solve(): extra arguments must be hashable Python values')
        fn = self.cache.get(key)
        if fn is None:
            params = [Tensor('param', attr=i, name=getattr(l, 'name', None)) for i, l in enumerate(leaves)]
            out = self.orig(_unflatten(struct, iter(params)), *args, **kw)
            oleaves = []
            ostruct = _flatten(out, oleaves)
            fn = self.cache[key] = StepFn(params, oleaves, ostruct, key)
        call = Call(fn, leaves)
        outs = [Tensor('apply', attr=(call, i)) for i in range(len(fn.outputs))]
        return _unflatten(fn.ostruct, iter(outs))


# ---------------------------------------------------------------------------------------------------
# programs: what one kernel mode computes
# ---------------------------------------------------------------------------------------------------
class Level:
    """one sub-step: new value of slot p = outs[p] evaluated with param i bound to slot bind[i]"""

    def __init__(self, fn, bind, outs):
        self.fn, self.bind, self.outs = fn, bind, outs

    def signature(self):
        return (id(self.fn), tuple(sorted(self.bind.items())), tuple(sorted((p, id(e)) for p, e in self.outs.items())))


class Program:
    """levels[k] = sub-step k of the op; mask = slots the op assigns"""

    def __init__(self, levels, mask, uses_lap, zeropad=False):
        self.levels, self.mask, self.uses_lap = levels, mask, uses_lap
        self.zeropad = zeropad            # the Laplacian is the zero-padded 3x3 convolution (fenton_simple.py:38-49)


def _walk(expr, seen, visit):
    """post-order over Tensors"""
    stack = [(expr, False)]
    while stack:
        node, done = stack.pop()
        if not isinstance(node, Tensor):
            continue
        if done:
            visit(node)
            continue
        if id(node) in seen:
            continue
        seen.add(id(node))
        stack.append((node, True))
        for a in node.args:
            stack.append((a, False))


class Compiler:
    def __init__(self, model):
        self.m = model
        self.grid = (model.height, model.width)
        self.slots = []                 # grid Variables in slab order
        self.host_vars = []             # small Variables kept on the host

    # ---- variables ---------------------------------------------------------------------------------
    def slot_of(self, var):
        if var._slot is None:
            if var.shape != self.grid:
                raise TraceError('variable %r has shape %s, the grid is %s' % (var.name, var.shape, self.grid))
            var._slot = len(self.slots)
            self.slots.append(var)
        return var._slot

    def _register_vars(self, expr):
        def visit(n):
            if n.op == 'var':
                if n.shape == self.grid:
                    self.slot_of(n)
                elif n not in self.host_vars:
                    self.host_vars.append(n)
        _walk(expr, set(), visit)

    # ---- one group -> Program + host assigns -------------------------------------------------------
    def split(self, group):
        grid, host = [], []
        for a in group.assigns:
            ref = a.ref
            if ref.op == 'index':
                host.append(a)
            elif ref.op == 'var' and ref.shape == self.grid:
                grid.append(a)
            else:
                raise TraceError('assign: %r is neither a grid variable nor an element of a host variable' % (ref,))
        return grid, host

    def program(self, assigns):
        if not assigns:
            return None
        applied = [isinstance(a.value, Tensor) and a.value.op == 'apply' for a in assigns]
        if all(applied):
            return self._chain(assigns)
        for a in assigns:
            self._check_no_apply(a.value)
        # no solve() wrapping: the group itself is the step function, its parameters are the variables
        outs = {}
        for a in assigns:
            self._register_vars(a.value)
            outs[self.slot_of(a.ref)] = a.value
        return self._finish([Level(None, {}, outs)], set(outs))

    def _check_no_apply(self, expr):
        def visit(n):
            if n.op == 'apply':
                raise TraceError('results of self.solve(...) must be assigned as they are (tf.assign(var, new_value)); '
                                 'arithmetic on them outside solve() is not traced')
        _walk(expr, set(), visit)

    def _chain(self, assigns):
        top = assigns[0].value.attr[0]
        for a in assigns:
            if a.value.attr[0] is not top:
                raise TraceError('one op must assign the results of ONE final solve(...) call')
        calls = [top]
        while True:
            c = calls[-1]
            kinds = set()
            below = None
            for x in c.inputs:
                if x.op == 'var':
                    kinds.add('var')
                elif x.op == 'apply':
                    kinds.add('apply')
                    if below is None:
                        below = x.attr[0]
                    elif x.attr[0] is not below:
                        raise TraceError('solve(state): the state must come from one previous solve(...) call')
                else:
                    raise TraceError('solve(state): state entries must be variables or the result of a previous '
                                     'solve(...), not expressions')
            if kinds == {'var'}:
                break
            if 'apply' not in kinds:
                raise TraceError('solve(state): empty state')
            # variables mixed into a later level are read-only parameters (never assigned): allowed
            calls.append(below)
        calls.reverse()                                     # bottom (first sub-step) first
        bottom = calls[0]
        slot_of_param = {}
        for i, v in enumerate(bottom.inputs):
            slot_of_param[i] = self.slot_of(v)
        if len(set(slot_of_param.values())) != len(slot_of_param):
            raise TraceError('solve(state): the same variable appears twice in the state')
        levels = []
        prev_out_slot = None                                # output index of the previous call -> slot it lands in
        for ci, c in enumerate(calls):
            bind = {}
            if ci == 0:
                bind = dict(slot_of_param)
            else:
                for i, x in enumerate(c.inputs):
                    if x.op == 'var':
                        bind[i] = self.slot_of(x)
                    else:
                        oi = x.attr[1]
                        if oi not in prev_out_slot:
                            raise TraceError('solve(): an output that is not part of the state feeds the next sub-step')
                        bind[i] = prev_out_slot[oi]
            # where do this call's outputs go?  They are the inputs of the next call (by position) or the
            # final assigns
            out_slot = {}
            if ci + 1 < len(calls):
                nxt = calls[ci + 1]
                # slot of the next call's param i = slot of the variable that param stands for at the bottom
                for i, x in enumerate(nxt.inputs):
                    if x.op == 'apply':
                        # param i of the next level carries the same state entry as param i of this level
                        if i not in bind:
                            raise TraceError('solve(): the state changes its layout between sub-steps')
                        out_slot[x.attr[1]] = bind[i]
            else:
                for a in assigns:
                    out_slot[a.value.attr[1]] = self.slot_of(a.ref)
            outs = {}
            for oi, p in out_slot.items():
                e = c.fn.outputs[oi]
                if p in outs:
                    raise TraceError('two outputs of solve() are assigned to the same variable')
                outs[p] = e
            for e in outs.values():
                self._register_vars(e)
            levels.append(Level(c.fn, bind, outs))
            prev_out_slot = out_slot
        mask = set(levels[-1].outs)
        if len(levels) > 1:
            carried = set()
            for lv in levels[:-1]:
                carried |= set(lv.outs)
            if not carried <= mask:
                raise TraceError('a chained tick must assign every variable its sub-steps update '
                                 '(missing: %s)' % sorted(self.slots[p].name or p for p in carried - mask))
        return self._finish(levels, mask)

    def _finish(self, levels, mask):
        uses_lap = zeropad = False
        for lv in levels:
            for e in lv.outs.values():
                found = []
                _walk(e, set(), lambda n: found.append(n) if n.op in ('lap', 'bnd', 'conv3') else None)
                for n in found:
                    self._note_pot(n, lv)
                    uses_lap = uses_lap or n.op in ('lap', 'conv3')
                    zeropad = zeropad or n.op == 'conv3'
                    if n.op == 'lap' and zeropad or n.op == 'conv3' and any(m.op == 'lap' for m in found):
                        raise TraceError('a model uses ONE Laplacian: IonicModel.laplace or the 3x3 convolution, not both')
        return Program(levels, mask, uses_lap, zeropad)

    pot_slot = None

    def _note_pot(self, node, lv):
        arg = node.args[0]
        if node.op == 'conv3':
            # the zero-padded 3x3 convolution of the stand-alone scripts (fenton_simple.py:38-49): the kernels carry
            # exactly one such stencil, the Laplacian [[.5, 1, .5], [1, -6, 1], [.5, 1, .5]] of an enforced array
            if tuple(node.attr) != (0.5, 1.0, 0.5, 1.0, -6.0, 1.0, 0.5, 1.0, 0.5):
                raise TraceError('tf.nn.depthwise_conv2d: the only 3x3 kernel the fused kernels implement is the '
                                 'Laplacian [[.5, 1, .5], [1, -6, 1], [.5, 1, .5]] (fenton_simple.py:46-48)')
            if not (isinstance(arg, Tensor) and arg.op == 'bnd'):
                raise TraceError('the 3x3 convolution must be applied to the boundary-enforced array '
                                 '(enforce_boundary(X), fenton_simple.py:131-135)')
            arg = arg.args[0]
        elif node.op == 'lap':
            if not (isinstance(arg, Tensor) and arg.op == 'bnd'):
                raise TraceError('self.laplace(X): X must be self.enforce_boundary(state variable) — the fused kernel '
                                 'implements the stencil with the no-flux boundary folded in (ionic.py:44-60,107-113)')
            arg = arg.args[0]
        if not isinstance(arg, Tensor) or arg.op not in ('param', 'var'):
            raise TraceError('self.enforce_boundary(X): X must be a state variable of the current sub-step')
        p = lv.bind[arg.attr] if arg.op == 'param' else self.slot_of(arg)
        if self.pot_slot is None:
            self.pot_slot = p
        elif self.pot_slot != p:
            raise TraceError('only one variable may diffuse (enter enforce_boundary / laplace)')


# ---------------------------------------------------------------------------------------------------
# code generation
# ---------------------------------------------------------------------------------------------------
def _lit(c):
    c = float(np.float32(c))
    if math.isnan(c):
        return '__builtin_nanf("")'
    if math.isinf(c):
        return '__builtin_inff()' if c > 0 else '(-__builtin_inff())'
    s = '%.9g' % c
    if 'e' not in s and '.' not in s:
        s += '.0'
    s += 'f'
    return '(%s)' % s if c < 0 or s.startswith('-') else s


# ---- division by a constant --------------------------------------------------------------------------------
# q = a*rc; r = fma(-q, c, a); fma(r, rc, q) with rc = RN(1/c) is the correctly rounded quotient for MOST constants
# (Markstein); the hand-written models carry an exhaustive check per constant (tools/ubench/divtest.c).  For a
# user's constants the same exhaustive check runs here, once per constant, over all 2^23 significands of `a`
# (the exponents of a and c only shift the result); a constant that fails — or that the float64 emulation cannot
# decide beyond doubt — keeps the true IEEE division under the rounding-faithful policy.
_DIVC = {}


def _divc_cache_file():
    return os.path.join(CACHE, 'divc_checked.json')


def _divc_load():
    if not _DIVC:
        try:
            with open(_divc_cache_file()) as f:
                _DIVC.update(json.load(f))
        except (OSError, ValueError):
            pass


def divc_is_exact(c):
    """True iff the 3-instruction form reproduces RN(a/c) for every float32 a (normal range)"""
    c32 = np.float32(c)
    key = c32.tobytes().hex()
    _divc_load()
    if key in _DIVC:
        return _DIVC[key]
    ok = _divc_check(abs(float(c32)))
    _DIVC[key] = ok
    try:
        os.makedirs(CACHE, exist_ok=True)
        tmp = '%s.%d.tmp' % (_divc_cache_file(), os.getpid())
        with open(tmp, 'w') as f:
            json.dump(_DIVC, f)
        os.replace(tmp, _divc_cache_file())
    except OSError:
        pass
    return ok


def _divc_check(c):
    if not np.isfinite(c) or c == 0.0:
        return False
    rc32 = np.float32(1.0 / c)                              # RN(1/c): double division then one rounding is safe (53 >= 2*24+2)
    if not np.isfinite(rc32) or abs(float(rc32)) < 1.2e-38 or c < 1.2e-38:
        return False
    rc = float(rc32)
    for lo in range(1 << 23, 1 << 24, 1 << 20):
        a = np.arange(lo, lo + (1 << 20), dtype=np.float64)                # every significand, exponent 2^23
        q = (a * rc).astype(np.float32).astype(np.float64)                  # a*rc is exact in double (48 bits)
        r = (a - q * c).astype(np.float32).astype(np.float64)               # exact difference, then fma's one rounding
        u = r * rc                                                          # exact (48 bits)
        s = q + u                                                           # TwoSum: s + e == q + u exactly
        bb = s - q
        e = (q - (s - bb)) + (u - bb)
        f = s.astype(np.float32)
        f64 = f.astype(np.float64)
        d = (s - f64) + e                                                   # exact sum minus the candidate rounding
        half = 0.5 * np.spacing(f)                                          # float32 ulp of the candidate
        half = np.where(np.abs(np.frexp(f64)[0]) == 0.5, 0.5 * half, half)  # below a power of two the ulp halves
        if np.any(np.abs(d) > half * (1 - 1e-9)):                           # wrong or too close to a tie to call
            return False
        want = (a / c).astype(np.float32)
        if not np.array_equal(f, want):
            return False
    return True


_TEMP = re.compile(r'^t\d+$')
_BINOPS = {'add': '+', 'sub': '-', 'mul': '*'}
_CMP = {'gt': 'vcmp_gt', 'ge': 'vcmp_ge', 'lt': 'vcmp_lt', 'le': 'vcmp_le', 'eq': 'vcmp_eq', 'ne': 'vcmp_ne'}
_UNARY = {'tanh': 'P::tanhv(%s)', 'exp': 'P::exp(%s)', 'expm1': 'P::expm1g(%s)', 'log': 'P::log(%s)',
          'sqrt': 'g_sqrt<P>(%s)', 'abs': 'g_abs(%s)', 'sign': 'g_sign(%s)', 'reciprocal': 'P::rcp(%s)',
          'neg': '-%s'}


class _Emitter:
    """one level -> C++ statements (post-order = the reference's evaluation order, one statement per node)"""

    def __init__(self, level, remap, pot, counter):
        self.lv, self.remap, self.pot = level, remap, pot
        self.names, self.lines, self.n = {}, [], counter
        # how many times each node is consumed: a multiply may only melt into P::mad if nothing else reads it
        self.uses = {}
        seen = set()

        def visit(n):
            for a in n.args:
                if isinstance(a, Tensor):
                    self.uses[id(a)] = self.uses.get(id(a), 0) + 1
        for e in level.outs.values():
            if isinstance(e, Tensor):
                self.uses[id(e)] = self.uses.get(id(e), 0) + 1
                _walk(e, seen, visit)

    # ---- peepholes: forms that are the SAME float32 function under the rounding-faithful policy and cheaper
    # ---- (or fusable) under the fast one -------------------------------------------------------------------
    def _single(self, x, op):
        return isinstance(x, Tensor) and x.op == op and self.uses.get(id(x), 0) == 1 and id(x) not in self.names

    def _heaviside(self, node):
        """(1 +- sign(x)) * 0.5  ->  g_heav / g_heav_not (fenton.py:73-79)"""
        if node.op != 'mul':
            return None
        a, b = node.args
        if not isinstance(a, Tensor):
            a, b = b, a
        if isinstance(b, Tensor) or b != 0.5 or not isinstance(a, Tensor) or a.op not in ('add', 'sub'):
            return None
        if self.uses.get(id(a), 0) != 1 or id(a) in self.names:
            return None
        p, q = a.args
        if a.op == 'add' and not isinstance(p, Tensor):
            p, q = q, p
        if a.op == 'add' and self._single(p, 'sign') and not isinstance(q, Tensor) and q == 1.0:
            return 'g_heav(%s)' % self.ref(p.args[0])
        if a.op == 'sub' and not isinstance(p, Tensor) and p == 1.0 and self._single(q, 'sign'):
            return 'g_heav_not(%s)' % self.ref(q.args[0])
        return None

    def _fused(self, node):
        """x*y + z, z - x*y, x*y - z with a multiply nobody else reads -> P::mad / P::mad3;  1 + tanh(x)"""
        op = node.op
        a, b = node.args
        if op == 'add':
            for t, o in ((a, b), (b, a)):
                if self._single(t, 'tanh') and not isinstance(o, Tensor) and o == 1.0:
                    return 'P::one_plus_tanh(%s)' % self.ref(t.args[0])
        cands = [(a, b, False, False), (b, a, False, False)] if op == 'add' else \
                [(b, a, True, False), (a, b, False, True)]          # z - x*y ;  x*y - z
        for m, z, neg_m, neg_z in cands:
            if not self._single(m, 'mul'):
                continue
            x, y = m.args
            if not isinstance(x, Tensor):
                x, y = y, x
            zs = self.ref(z)
            if neg_z:
                zs = '-%s' % zs if isinstance(z, Tensor) else _lit(-z)
            if not isinstance(y, Tensor):
                return 'P::mad(%s, %s, %s)' % (self.ref(x), _lit(-y if neg_m else y), zs)
            if isinstance(z, Tensor):
                xs = self.ref(x)
                return 'P::mad3(%s, %s, %s)' % ('-%s' % xs if neg_m else xs, self.ref(y), zs)
        return None

    def new(self, prefix='t'):
        self.n[0] += 1
        return '%s%d' % (prefix, self.n[0])

    def ref(self, x):
        return self.emit(x) if isinstance(x, Tensor) else _lit(x)

    def as_T(self, x):
        return self.emit(x) if isinstance(x, Tensor) else 'T_of<T>(%s)' % _lit(x)

    def slot_expr(self, node):
        if node.op == 'var' and node._slot is None:
            raise TraceError('variable %r is not a grid array: small host variables cannot enter the cell arithmetic'
                             % (node.name,))
        p = self.lv.bind[node.attr] if node.op == 'param' else node._slot
        return 's[%d]' % self.remap[p]                      # (generation runs before slots are renumbered)

    def emit(self, node):
        k = id(node)
        if k in self.names:
            return self.names[k]
        op, a = node.op, node.args
        if op in ('param', 'var'):
            if op == 'param' and node.attr not in self.lv.bind:
                raise TraceError('solve(): a state entry that was never bound is used')
            r = self.slot_expr(node)
        elif op == 'bnd':
            r = 'V0'
        elif op in ('lap', 'conv3'):
            r = 'lap'
        elif op in _BINOPS:
            special = self._heaviside(node) if op == 'mul' else self._fused(node)
            r = self.stmt(special if special else '%s %s %s' % (self.ref(a[0]), _BINOPS[op], self.ref(a[1])))
        elif op == 'div':
            if not isinstance(a[1], Tensor):
                if divc_is_exact(a[1]):                     # x / c through RN(1/c): 3 instructions, bit-identical
                    rc = float(np.float32(1.0) / np.float32(a[1]))
                    r = self.stmt('P::divc(%s, %s, %s)' % (self.ref(a[0]), _lit(a[1]), _lit(rc)))
                else:
                    r = self.stmt('P::divk(%s, %s)' % (self.ref(a[0]), _lit(a[1])))
            else:
                r = self.stmt('P::div(%s, %s)' % (self.ref(a[0]), self.ref(a[1])))
        elif op in _CMP:
            r = self.stmt('%s(%s, %s)' % (_CMP[op], self.ref(a[0]), self.ref(a[1])), mask=True)
        elif op in ('and', 'or'):
            r = self.stmt('vm_%s(%s, %s)' % (op, self.emit(a[0]), self.emit(a[1])), mask=True)
        elif op == 'not':
            r = self.stmt('vm_not(%s)' % self.emit(a[0]), mask=True)
        elif op == 'where':
            r = self.stmt('vsel(%s, %s, %s)' % (self.emit(a[0]), self.as_T(a[1]), self.as_T(a[2])))
        elif op == 'square':
            x = self.ref(a[0])
            r = self.stmt('%s * %s' % (x, x))
        elif op == 'pow':
            x, e = self.ref(a[0]), a[1]
            if e == 2.0:
                r = self.stmt('%s * %s' % (x, x))
            elif e == 3.0:                                  # as the stock Courtemanche kernel: within 1 ulp of powf
                r = self.stmt('(%s * %s) * %s' % (x, x, x))
            elif e == 1.0:
                r = x
            else:
                r = self.stmt('g_pow(%s, %s)' % (x, _lit(e)))
        elif op in ('maximum', 'minimum'):
            r = self.stmt('g_%s(%s, %s)' % (op[:3], self.ref(a[0]), self.ref(a[1])))
        elif op == 'clip':
            if isinstance(a[1], Tensor) or isinstance(a[2], Tensor):
                r = self.stmt('g_min(g_max(%s, %s), %s)' % (self.ref(a[0]), self.ref(a[1]), self.ref(a[2])))
            else:
                r = self.stmt('clipf(%s, %s, %s)' % (self.ref(a[0]), _lit(a[1]), _lit(a[2])))
        elif op in _UNARY:
            r = self.stmt(_UNARY[op] % self.ref(a[0]))
        elif op == 'apply':
            raise TraceError('a solve() result is used inside another expression')
        elif op == 'index':
            raise TraceError('element access V[i, j] is only available as the source of a probe assign')
        else:
            raise TraceError('op %r is not supported by the code generator' % op)
        self.names[k] = r
        return r

    def stmt(self, rhs, mask=False):
        name = self.new('m' if mask else 't')
        self.lines.append('const %s %s = %s;' % ('auto' if mask else 'T', name, rhs))
        return name


_HEAVY = frozenset(('tanh', 'exp', 'expm1', 'log', 'sqrt', 'div', 'reciprocal', 'pow'))


def _weight(level):
    """rough instruction count of one sub-step: graph nodes, transcendental / division nodes counted 4x"""
    total = [0]

    def visit(node):
        if node.op not in ('param', 'var', 'bnd', 'lap', 'conv3'):
            total[0] += 4 if node.op in _HEAVY else 1
    seen = set()
    for e in level.outs.values():
        _walk(e, seen, visit)
    return total[0]


def generate(programs, nslots, remap, spt):
    """programs[mode] -> text of the generated header"""
    pot = 0
    n = nslots
    tick = programs[0]
    K = len(tick.levels)
    zeropad = any(getattr(p, 'zeropad', False) for p in programs)
    # (the zero-padded tap rule lives in tick_kernel only: such a model runs one sub-step per launch)
    fuse = 1 < K <= 12 and K == spt and not zeropad
    R = 3 if n <= 5 else 2
    TX, TY = 64 - 2 * K, 16 * R - 2 * K
    if fuse and (TX < 16 or TY < 6):
        fuse = False
    # Small grids are launch/latency-bound: the whole tick in one launch.  Large grids are throughput-bound, where
    # the redundant rim of a deep fusion costs more than launches: a cheap model keeps a shallow
    # fusion (K2 <= 5), a heavy one goes back to one sub-step per launch.  "cheap" = few graph nodes per sub-step.
    weight = max(_weight(lv) for lv in tick.levels)
    K2 = 1
    if fuse and weight < 250:
        K2 = max(d for d in range(1, 6) if K % d == 0)
    # (a few state arrays: 12 full strips of 3 rows — the waves of a workgroup spread evenly over the four SIMDs —
    # measured best for the hand-written 4-variable model at 1024^2 .. 4096^2)
    R2 = 3 if n <= 5 else 2
    TX2, TY2 = 64 - 2 * K2, (36 - 2 * (K2 - 1)) if n <= 5 else 8 * R2 - 2 * K2
    out = ['// generated by fib_tf_amd/traced.py from a traced model graph — do not edit',
           '// graph weight per sub-step: %d' % weight,
           '#define FIB_CUSTOM_K %d' % (K if fuse else 1),
           '#define FIB_CUSTOM_TX %d' % (TX if fuse else 64),
           '#define FIB_CUSTOM_TY %d' % (TY if fuse else 4),
           '#define FIB_CUSTOM_R %d' % R,
           '#define FIB_CUSTOM_TYB %d' % ((15 * R - 2 * K) if fuse and 15 * R - 2 * K >= 6 else 0),   # one wave fewer per tile
           '#define FIB_CUSTOM_K2 %d' % K2,
           '#define FIB_CUSTOM_TX2 %d' % TX2,
           '#define FIB_CUSTOM_TY2 %d' % TY2,
           '#define FIB_CUSTOM_R2 %d' % R2,
           'struct Custom {',
           '    static constexpr int NVAR = %d;' % n,
           '    static constexpr int DEFAULT_STEPS = %d;' % spt,
           '    static constexpr int NMODES = %d;' % len(programs),
           '    static constexpr bool HAS_VEC = true;'] + (
           ['    static constexpr bool ZEROPAD = true;    // taps outside the grid read 0 (3x3 convolution, padding SAME)']
           if zeropad else []) + [
           '    struct Consts { float unused; };',
           '    template <class C> static FIB_DEV const C &pinned(const C &k) { return k; }',
           '    static constexpr unsigned mask(int mode)', '    {']
    for mode, prog in enumerate(programs):
        bits = 0
        for p in prog.mask:
            bits |= 1 << remap[p]
        out.append('        if (mode == %d) return 0x%Xu;' % (mode, bits))
    out += ['        return 0u;', '    }',
            '    template <class P, int MODE>',
            '    static FIB_DEV void step(float (&s)[NVAR], float V0, float lap, const Consts &, int sub)',
            '    {', '        body<P, MODE, float>(s, V0, lap, sub);', '    }',
            '    template <class P, int MODE, int R>',
            '    static FIB_DEV void stepN(float (&s)[R][NVAR], const float (&V0)[R], const float (&lap)[R], const Consts &, int sub)',
            '    {',
            '        vf<R> t[NVAR], v0, l;',
            '        _Pragma("unroll") for (int r = 0; r < R; ++r) {',
            '            _Pragma("unroll") for (int v = 0; v < NVAR; ++v) t[v].v[r] = s[r][v];',
            '            v0.v[r] = V0[r];', '            l.v[r] = lap[r];', '        }',
            '        body<P, MODE, vf<R>>(t, v0, l, sub);',
            '        _Pragma("unroll") for (int r = 0; r < R; ++r)',
            '            _Pragma("unroll") for (int v = 0; v < NVAR; ++v) s[r][v] = t[v].v[r];',
            '    }',
            '    template <class P, int MODE, class T>',
            '    static FIB_DEV void body(T (&s)[NVAR], const T &V0, const T &lap, int sub)',
            '    {']
    counter = [0]
    for mode, prog in enumerate(programs):
        out.append('        %sif constexpr (MODE == %d) {' % ('' if mode == 0 else 'else ', mode))
        # distinct sub-step kinds
        kinds, order = {}, []
        for lv in prog.levels:
            sig = lv.signature()
            if sig not in kinds:
                kinds[sig] = len(kinds)
            order.append(kinds[sig])
        emitted = {}
        for lv, kind in zip(prog.levels, order):
            if kind in emitted:
                continue
            em = _Emitter(lv, remap, pot, counter)
            results = [(remap[p], em.ref(e)) for p, e in sorted(lv.outs.items(), key=lambda kv: remap[kv[0]])]
            body = list(em.lines)
            # every read of s[] happens above; results that are not fresh temporaries (a passed-through
            # variable, V0, a constant) are latched first, then all stores
            stores = []
            for slot, r in results:
                if not _TEMP.match(r):
                    nm = em.new('n')
                    rhs = r if (r.startswith('s[') or r in ('V0', 'lap')) else 'T_of<T>(%s)' % r
                    body.append('const T %s = %s;' % (nm, rhs))
                    r = nm
                stores.append('s[%d] = %s;' % (slot, r))
            body += stores
            emitted[kind] = body
        if len(emitted) == 1:
            out += ['            ' + l for l in emitted[0]]
        else:
            subs_of = {}
            for sub, kind in enumerate(order):
                subs_of.setdefault(kind, []).append(sub)
            first = True
            for kind, body in emitted.items():
                cond = ' || '.join('sub == %d' % s_ for s_ in subs_of[kind])
                out.append('            %sif (%s) {' % ('' if first else 'else ', cond))
                out += ['                ' + l for l in body]
                out.append('            }')
                first = False
        out.append('        }')
    out += ['        (void)sub; (void)V0; (void)lap;', '    }', '};', '']
    return '\n'.join(out)


# ---------------------------------------------------------------------------------------------------
# the model base class
# ---------------------------------------------------------------------------------------------------
class IonicModel(_HostModel):
    """drop-in for the reference's `ionic.IonicModel` when the model file's `tf` is fib_tf_amd.tfgraph"""

    MODEL_ID = _lib.CUSTOM

    def __init__(self, config):
        super().__init__(config)
        self._State = {}
        self._ode_op = None
        self._compiled = None
        self._library = None
        self._modes = {}                # id(Group) -> (mode index or None, host assigns)

    # ---- building blocks: symbolic on tensors, device ops on arrays ---------------------------------
    def laplace(self, X0):
        if isinstance(X0, Tensor):
            return Tensor('lap', (X0,))
        return super().laplace(X0)

    def enforce_boundary(self, X):
        if isinstance(X, Tensor):
            return Tensor('bnd', (X,))
        return super().enforce_boundary(X)

    def phase_field(self, X):
        if isinstance(X, Tensor):
            raise TraceError('phase_field is part of self.laplace in the fused kernel; call self.laplace')
        return super().phase_field(X)

    def rush_larsen(self, g, g_inf, g_tau, dt, name=None):
        if any(isinstance(x, Tensor) for x in (g, g_inf, g_tau)):
            # ionic.py:115-123
            return tf.clip_by_value(g + (g - g_inf) * tf.expm1(-dt / g_tau), 1e-5, 0.99999)
        return super().rush_larsen(g, g_inf, g_tau, dt, name)

    def jit_scope(self):
        return self                                         # the reference's no-op fallback, ionic.py:299-307

    # ---- define(): called first thing by the subclass's define() -----------------------------------
    def define(self, s1=True):
        super().define(s1)
        solve = getattr(type(self), 'solve', None)
        if solve is not None and not isinstance(self.__dict__.get('solve'), _SolveTracer):
            self.solve = _SolveTracer(self, solve.__get__(self))

    # ---- compile on first use -------------------------------------------------------------------------
    def _ensure_compiled(self):
        if self._stepper is not None and self._compiled is not None:
            return
        c = self._analyze()
        self._library = _build(c['source'], device=self.device)
        for v in c['slots']:
            v._owner = self
        for v in c['host_vars']:
            v._owner, v._host = self, v.init.copy()
        self._create_traced([v.init for v in c['slots']], c['spt'])

    def _analyze(self):
        """graph -> programs + generated source (pure Python: no compiler, no device)"""
        if self._compiled is not None:
            return self._compiled
        if not self.defined:
            raise AssertionError('run should be called after calling define')
        if not isinstance(self._ode_op, (Group, Assign)):
            raise TraceError('define() must set self._ode_op = tf.group(<assign ops>)')
        comp = Compiler(self)
        if isinstance(self._ode_op, Assign):
            self._ode_op = Group((self._ode_op,))
        groups = [self._ode_op]
        names = ['_ode_op']
        for name, op in self._ops.items():
            if isinstance(op, Assign):
                op = self._ops[name] = Group((op,))
            if isinstance(op, Group):
                groups.append(op)
                names.append(name)
        programs, table = [], {}
        for g, name in zip(groups, names):
            grid, host = comp.split(g)
            prog = comp.program(grid)
            mode = None
            if prog is not None:
                if name != '_ode_op':
                    if prog.uses_lap:
                        raise TraceError("op %r uses the Laplacian: only the tick op (_ode_op) may" % name)
                    if len(prog.levels) != 1:
                        raise TraceError('op %r chains several solve() calls: only the tick op may' % name)
                mode = len(programs)
                programs.append((name, prog))
            elif name == '_ode_op':
                raise TraceError('_ode_op assigns no grid variable')
            table[id(g)] = (mode, host)
        if len(programs) > MAX_MODES:
            raise TraceError('at most %d assign groups' % MAX_MODES)
        for _, host in table.values():
            for a in host:
                self._check_probe(comp, a)
        n = len(comp.slots)
        if n > MAX_VARS:
            raise TraceError('at most %d state variables' % MAX_VARS)
        pot = comp.pot_slot if comp.pot_slot is not None else 0
        for name, prog in programs[1:]:
            if pot in prog.mask:
                raise TraceError('op %r assigns the diffusing variable: only the tick op may' % name)
        if pot not in programs[0][1].mask and programs[0][1].uses_lap:
            raise TraceError('the tick op uses the Laplacian but does not assign the diffusing variable')
        # slab order: the diffusing variable first (the kernels' variable 0)
        order = [pot] + [p for p in range(n) if p != pot]
        remap = {p: i for i, p in enumerate(order)}
        spt = len(programs[0][1].levels)
        want = int(getattr(self, 'dt_per_step', spt) or spt)
        if want != spt:
            raise TraceError('define() set dt_per_step = %d but the tick op chains %d solve() calls' % (want, spt))
        src = generate([p for _, p in programs], n, remap, spt)
        self._modes = table
        # from here on a variable's slot is its index in the device slab; programs keep compile-time slots
        # and go through `remap`
        for v in comp.slots:
            v._cslot, v._slot = v._slot, remap[v._slot]
        slots = sorted(comp.slots, key=lambda v: v._slot)
        # names: tf.Variable(name=...), else the key under which define() filed the variable in self._State
        # (court.py:87-90,106), else positional
        by_id = {id(v): k for k, v in self._State.items()} if isinstance(self._State, dict) else {}
        for i, v in enumerate(slots):
            if v.name is None:
                v.name = by_id.get(id(v), 'var%d' % i)
        self.VAR_NAMES = tuple(v.name for v in slots)
        self._compiled = {'source': src, 'programs': programs, 'remap': remap, 'slots': slots,
                          'host_vars': comp.host_vars, 'spt': spt, 'pot': pot, 'modes': table}
        return self._compiled

    def _check_probe(self, comp, a):
        ref, val = a.ref, a.value
        tgt = ref.args[0]
        if not isinstance(tgt, Variable) or tgt.shape == comp.grid:
            raise TraceError('element assigns are for small host variables (the reference\'s Trend probe)')
        if not (isinstance(val, Tensor) and val.op == 'index' and isinstance(val.args[0], Variable)
                and val.args[0].shape == comp.grid):
            raise TraceError('an element of a host variable can only be assigned one cell of a grid variable: '
                             'tf.assign(Trend[0], V[row, col])')
        comp.slot_of(val.args[0])
        if tgt not in comp.host_vars:
            comp.host_vars.append(tgt)

    def _new_stepper(self, steps_per_tick=0, shard=True):
        from .sharded import ShardedStepper, dist_world
        _, world = dist_world()
        if shard and world > 1:
            st = ShardedStepper(self.MODEL_ID, self.height, self.width, self.dt, self.diff, flags=self._flags(),
                                device=self.device, steps_per_tick=steps_per_tick,
                                halo_ticks=getattr(self, 'halo_ticks', 4), library=self._library)
        else:
            st = _lib.Stepper(self.MODEL_ID, self.height, self.width, self.dt, self.diff, flags=self._flags(),
                              device=self.device, steps_per_tick=steps_per_tick, library=self._library)
        if self.phase is not None:
            st.set_phase(self.phase)
        return st

    def _create_traced(self, init_arrays, spt):
        if self._stepper is not None:
            self._stepper.close()
        st = self._new_stepper(spt)
        st.set_state(-1, np.stack([np.asarray(a, np.float32) for a in init_arrays]))
        self._stepper = st
        self.dt_per_step = st.steps_per_tick
        if not self._State:
            self._State = {v.name or 'var%d' % v._slot: v for v in self._compiled['slots']}

    # ---- execution -------------------------------------------------------------------------------------
    def _eval_variable(self, var):
        if var._host is not None:
            return var._host.copy()
        return self._stepper.get_state(var._slot)

    def _run_group(self, g):
        self._ensure_compiled()
        if g is self._ode_op:
            self._stepper.step(1)
            return
        entry = self._modes.get(id(g))
        if entry is None:
            raise TraceError('this op was created after the model was compiled (register it in define() or via '
                             '_ops before the first run)')
        mode, host = entry
        if mode == 0:
            self._stepper.step(1)
        elif mode is not None:
            self._stepper.step_mode(mode)
        for a in host:
            src = a.value
            r, c = src.attr
            val = self._stepper.probe(src.args[0]._slot, int(r), int(c))
            a.ref.args[0]._host[a.ref.attr] = val

    def fire_op(self, name):
        op = self._ops[name]
        if isinstance(op, (Group, Assign)):
            self._ensure_compiled()
            self._run_group(self._ops[name])
        else:
            self._ensure_compiled()
            super().fire_op(name)

    def ode_op(self, tick):
        return self._ode_op

    def run(self, im=None, keep_state=False, block=True):
        self._ensure_compiled()
        yield from super().run(im, keep_state, block)

    def generated_source(self):
        """the HIP source emitted for this model (after define()); needs neither hipcc nor a GPU"""
        return self._analyze()['source']


def _source_hash(src):
    h = hashlib.sha1(src.encode())
    for d in _lib.DEPS:
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(_lib.HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _header_facts(src):
    """what the host needs to know about a generated header: sizes, assign masks, plan hints (parsed from its text)"""
    def num(pat):
        return int(re.search(pat, src).group(1), 0)
    facts = {'nvar': num(r'NVAR = (\d+);'), 'spt': num(r'DEFAULT_STEPS = (\d+);'), 'nmodes': num(r'NMODES = (\d+);'),
             'consts_bytes': 4,
             'plan': {k: num(r'#define FIB_CUSTOM_%s (\d+)' % k) for k in ('K', 'TX', 'TY', 'R', 'TYB', 'K2', 'TX2', 'TY2', 'R2')}}
    masks = [0] * 8
    for m in re.finditer(r'if \(mode == (\d+)\) return (0x[0-9A-Fa-f]+)u;', src):
        masks[int(m.group(1))] = int(m.group(2), 16)
    facts['masks'] = masks
    return facts


def _module_kernels(facts):
    """the kernels a traced model needs, as csrc/fibhip.hip lists them for a per-model build: (name expression, meta)"""
    p = facts['plan']
    out = []
    for fast, pol in ((0, 'fib::Exact'), (1, 'fib::Fast')):
        for phase in (0, 1):
            ph = 'true' if phase else 'false'
            shapes = [(0, 1, 64, 4, 256)]                                    # kind, K, TX, TY, NT
            if p['K'] > 1:
                shapes.append((1, p['K'], p['TX'], p['TY'], -p['R']))
                if p['TYB'] > 0:
                    shapes.append((1, p['K'], p['TX'], p['TYB'], -p['R']))
            if p['K2'] > 1 and p['K2'] != p['K']:
                shapes.append((1, p['K2'], p['TX2'], p['TY2'], -p['R2']))
            for kind, K, TX, TY, NT in shapes:
                fn = 'tick_kernel' if kind == 0 else 'strip_kernel'
                expr = 'fib::%s<fib::Custom, %s, 0, %d, %d, %d, %d, %s>' % (fn, pol, K, TX, TY, NT if kind == 0 else -NT, ph)
                out.append((expr, {'kind': kind, 'mode': 0, 'fast': fast, 'phase': phase, 'K': K, 'TX': TX, 'TY': TY, 'NT': NT}))
                # a strip that fuses the WHOLE tick also exists as the multi-tick launch (kind 3, listed right behind its
                # strip kernel): up to 32 ticks per launch on grids whose tiles are all resident at once (csrc/kernels.hpp)
                if kind == 1 and K == facts['spt'] and TX + 2 * (K - 1) == 62 and TX >= K and TY >= K:
                    expr = 'fib::strip_mt_kernel<fib::Custom, %s, 0, %d, %d, %d, %d, %s>' % (pol, K, TX, TY, -NT, ph)
                    out.append((expr, {'kind': 3, 'mode': 0, 'fast': fast, 'phase': phase, 'K': K, 'TX': TX, 'TY': TY, 'NT': NT}))
        for mode in range(1, facts['nmodes']):
            expr = 'fib::pointwise_kernel<fib::Custom, %s, %d>' % (pol, mode)
            out.append((expr, {'kind': 2, 'mode': mode, 'fast': fast, 'phase': 0, 'K': 1, 'TX': 0, 'TY': 0, 'NT': 0}))
    return out


_modules = {}


def _build(src, verbose=False, device=None):
    """generated header -> something `_lib.Stepper(library=...)` can run the model on.

    Default: compiled IN THIS PROCESS by hiprtc into a code object (cached as _traced/model_<hash>.hsaco + .json on the
    hash of the header and of csrc/) and loaded into the stock library (`fibhip_module_load`): no compiler on the box,
    no library per model.  FIBTF_TRACED_BUILD=hipcc (or a runtime without libhiprtc): the translation unit built by
    hipcc into _traced/libfibhip_<hash>.so, as before.  device=None compiles and caches only (no GPU needed)."""
    os.makedirs(CACHE, exist_ok=True)
    tag = _source_hash(src)
    how = os.environ.get('FIBTF_TRACED_BUILD', 'hiprtc')
    if how == 'hiprtc' and _lib.hiprtc() is not None:
        hsaco = os.path.join(CACHE, 'model_%s.hsaco' % tag)
        meta_path = os.path.join(CACHE, 'model_%s.json' % tag)
        if not (os.path.exists(hsaco) and os.path.exists(meta_path)):
            facts = _header_facts(src)
            kernels = _module_kernels(facts)
            code, lowered = _lib.rtc_compile('#include "kernels.hpp"\n', {'fib_custom_model.inc': src}, [e for e, _ in kernels],
                                             _lib.RTC_OPTIONS + ['-DFIB_CUSTOM_MODEL_INC="fib_custom_model.inc"'])
            facts['kernels'] = [dict(m, symbol=lowered[e]) for e, m in kernels]
            for path, data, mode in ((hsaco, code, 'wb'), (meta_path, json.dumps(facts), 'w')):
                tmp = '%s.%d.tmp' % (path, os.getpid())                 # atomic: several ranks may build the same model
                with open(tmp, mode) as f:
                    f.write(data)
                os.replace(tmp, path)
            if verbose:
                print('hiprtc: %d kernels, %d bytes -> %s' % (len(kernels), len(code), os.path.relpath(hsaco)))
        if device is None:
            return None
        key = (tag, device)
        if key not in _modules:
            with open(hsaco, 'rb') as f:
                code = f.read()
            with open(meta_path) as f:
                meta = json.load(f)
            _modules[key] = _lib.ModuleLibrary(code, meta, device, os.path.basename(hsaco))
        return _modules[key]
    inc = os.path.join(CACHE, 'model_%s.inc' % tag)
    so = os.path.join(CACHE, 'libfibhip_%s.so' % tag)
    if not os.path.exists(so):
        tmp = '%s.%d.tmp' % (inc, os.getpid())
        with open(tmp, 'w') as f:
            f.write(src)
        os.replace(tmp, inc)
        _lib.build_custom(inc, so, verbose=verbose)
    if device is None:
        return None
    _lib.warm(device)                                 # the stock library's kernels first (include/fibhip.h fibhip_warm)
    return _lib.load(so)
