"""fenton_simple — the stand-alone Fenton 4v scripts of the reference (siravan/fib_tf `fenton_simple.py:59-221`;
`fenton_jit.py` is the same model inside an XLA scope): no IonicModel base class, one `solve` per op, the Laplacian
written as a zero-padded 3x3 convolution (`fenton_simple.py:38-49`) instead of `IonicModel.laplace`, its own S2 op
(`U = max(U, s2_init)` on the upper-left quadrant INCLUDING row 0 / column 0, `:150-152,172`) and its own `run(im)`
that is not a generator (`:175-221`).

Device side: the stock Fenton kinetics with `FIBHIP_ZEROPAD` (csrc/kernels.hpp: taps outside the grid read 0; only the
outermost ring of cells differs from `Fenton4v` — `enforce_boundary` overwrites it at the top of every step, so the
interior evolves identically).  Steps between two host events (S2, a frame) are fused up to 10 per launch, exactly
like `Fenton4v`'s tick."""
import math
import os
import time

import numpy as np

from . import _lib


class Fenton4vSimple:
    def __init__(self, props):
        for key, val in props.items():
            setattr(self, key, val)
        self.min_v = 0.0
        self.max_v = 1.0
        self._stepper = None
        for key, default in (('device', int(os.environ.get('LOCAL_RANK', '0'))), ('fast_math', True),
                             ('timeline_name', 'timeline_simple.json')):
            if not hasattr(self, key):
                setattr(self, key, default)

    def _chunk(self, with_frames):
        """sub-steps per launch: the host only acts after step s2_step and (with a screen) every dt_per_plot steps"""
        g = math.gcd(int(self.samples), self._s2_step + 1)
        if with_frames:
            g = math.gcd(g, int(self.dt_per_plot))
        return max(d for d in range(1, 11) if g % d == 0)

    def define(self):
        """initial state u=min_v, v=w=1, s=0, S1 `u[:,1] = max_v` (fenton_simple.py:140-148)"""
        H, W = self.height, self.width
        init = np.stack([np.full([H, W], self.min_v, np.float32), np.ones([H, W], np.float32),
                         np.ones([H, W], np.float32), np.zeros([H, W], np.float32)])
        init[0][:, 1] = self.max_v
        self._init = init
        self._s2_step = int(self.s2_time / self.dt)             # fenton_simple.py:191
        self._open(self._chunk(False))

    def _open(self, spt):
        if self._stepper is not None:
            state = self._stepper.get_state(-1)
            self._stepper.close()
        else:
            state = self._init
        flags = _lib.ZEROPAD | (_lib.FAST if self.fast_math else 0)
        self._stepper = _lib.Stepper(_lib.FENTON4V, self.height, self.width, self.dt, self.diff, flags=flags,
                                     device=self.device, steps_per_tick=spt)
        self._stepper.set_state(-1, state)
        self._spt = spt

    def eval(self):
        return self._stepper.get_state(0)

    def state(self):
        return self._stepper.get_state(-1)

    def fire_s2(self):
        """`sess.run(self._s2_op)`: U = max(U, s2_init), s2_init = max_v on [:H//2, :W//2], min_v elsewhere"""
        self._stepper.pace(0, self.height // 2, 0, self.width // 2, float(self.max_v), float(self.min_v))

    def run(self, im=None):
        """the reference's loop (fenton_simple.py:186-199): `samples` steps, S2 after step int(s2_time/dt), a frame of
        the raw potential every dt_per_plot steps; prints the elapsed time; writes a one-event Chrome trace"""
        if self._stepper is None:
            raise AssertionError('run should be called after calling define')
        spt = self._chunk(im is not None)
        if spt != self._spt:
            self._open(spt)
        st = self._stepper
        then = time.time()
        i = 0                                                   # index of the next step to run
        while i < self.samples:
            st.step(1)
            i += spt
            last = i - 1                                        # the reference acts after step `last`
            if last == self._s2_step:
                self.fire_s2()
            if im and last % self.dt_per_plot == 0:
                im.imshow(self.eval())
        st.sync()
        self.elapsed = time.time() - then
        # The reference traces ONE MORE sess.run(_ode_op) for its timeline file (fenton_simple.py:201-209), which also
        # advances its state by a step; here the state stays at `samples` steps and the event is the mean launch.
        launches = max(1, -(-int(self.samples) // spt))
        with open(self.timeline_name, 'w') as f:
            f.write('{"traceEvents": [{"name": "fibhip launch (%d steps fused)", "ph": "X", "ts": 0, "dur": %.3f, '
                    '"pid": 0, "tid": 0}]}' % (spt, self.elapsed * 1e6 / launches))
        print('elapsed: %f sec' % self.elapsed)
        if im:
            im.wait()


class Fenton4vJIT(Fenton4vSimple):
    """the reference's `fenton_jit.py`: the same model with `solve` inside an XLA JIT scope (`fenton_jit.py:128-135`)
    and another trace-file name.  Fusion is what the kernel does already, so only the name differs here."""

    def __init__(self, props):
        super().__init__(props)
        if 'timeline_name' not in props:
            self.timeline_name = 'timeline_jit.json'
