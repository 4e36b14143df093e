"""IonicModel — the reference's config-dict / define() / run() stepper API (siravan/fib_tf
`ionic.py:30-307`) on top of libfibhip.so.

Driver code written for the reference runs unchanged against this module:

    model = Fenton4v(config)
    model.add_hole_to_phase_field(256, 256, 30)
    model.define()
    model.add_pace_op('s2', 'luq', 1.0)
    for i in model.run(im):
        if i == s2:
            model.fire_op('s2')

What differs underneath: `define()` does not build a TensorFlow graph, it creates a fibhip
handle whose state lives as one SoA slab in HBM; one `run()` tick is one call of `fibhip_step`
(one fused stencil+reaction launch per tick or per few sub-steps) instead of `Session.run`.
All numerical work happens in the HIP library; there is no CPU fallback here.
"""
import json
import os
import time

import numpy as np

from . import _lib


class StateVar:
    """stand-in for the `tf.Variable` handles the reference hands out (`pot()`, `_State[s]`):
    `.eval()` copies the array from the device (ionic.py:229, fenton.py:153)."""

    def __init__(self, model, index, name):
        self._model, self.index, self.name = model, index, name

    def eval(self):
        return self._model._stepper.get_state(self.index)

    def __array__(self, dtype=None, copy=None):
        a = self.eval()
        return a if dtype is None else a.astype(dtype)


class IonicModel:
    """base class for cardiac electrophysiology simulation (ionic.py:30-42)"""

    MODEL_ID = None
    VAR_NAMES = ()

    def __init__(self, config):
        for key, val in config.items():         # every key becomes an attribute, ionic.py:35-37
            setattr(self, key, val)
        self.phase = None
        self._ops = {}
        self.defined = False
        self.dt_per_step = 1
        self.cl_observer = None
        self._stepper = None
        self._library = None            # a specialised build of libfibhip (None = the stock library)
        # config['fast_math'] (default True): hardware exp/log/rcp/sqrt and reciprocal-multiply for the
        # model constants; False selects the rounding-faithful policy (IEEE-equivalent division, ocml
        # expf/tanhf/expm1f/logf): one float32 rounding per reference op.  Both are parity-tested; the
        # stencil and the phase-field term are bit-identical to the reference under either.
        for key, default in (('timeline', False), ('timeline_name', 'timeline.json'), ('save_graph', False),
                             ('device', int(os.environ.get('LOCAL_RANK', '0'))), ('fast_math', True)):
            if not hasattr(self, key):
                setattr(self, key, default)

    # ---- building blocks, callable on host arrays (ionic.py:44-123); executed on the GPU -------
    def laplace(self, X0):
        """9-point Laplacian of X0 (+ phase-field correction when self.phase is set), ionic.py:44-60"""
        return _lib.unit_op(1, X0, phi=self.phase, fast=self.fast_math, device=self.device)

    def phase_field(self, X):
        """phase-field correction for the REFLECT-padded array X, ionic.py:70-81"""
        X = np.asarray(X, np.float32)
        return _lib.unit_op(2, X[1:-1, 1:-1], phi=self.phase, fast=self.fast_math, device=self.device)

    def enforce_boundary(self, X):
        """no-flux (Neumann) boundary: interior re-padded SYMMETRIC, ionic.py:107-113"""
        return _lib.unit_op(0, X, device=self.device)

    def rush_larsen(self, g, g_inf, g_tau, dt, name=None):
        """clip(g + (g - g_inf) * expm1(-dt/tau), 1e-5, 0.99999), ionic.py:115-123"""
        g = np.asarray(g, np.float32)
        gi = np.broadcast_to(np.asarray(g_inf, np.float32), g.shape)
        gt = np.broadcast_to(np.asarray(g_tau, np.float32), g.shape)
        n = g.size
        rows = max(3, -(-n // 3))                # the device op works on a [rows, 3] sheet

        def sheet(a):
            buf = np.ones(rows * 3, np.float32)
            buf[:n] = np.ravel(a)
            return buf.reshape(rows, 3)

        out = _lib.unit_op(3, sheet(g), sheet(gi), sheet(gt), dt=dt, fast=self.fast_math, device=self.device)
        return out.ravel()[:n].reshape(g.shape)

    # ---- geometry (host side, as in the reference) ---------------------------------------------
    def add_hole_to_phase_field(self, x, y, radius, neg=False):
        """multiplies a circular hole centred at column x, row y into the phase field; with
        neg=True the disc is kept and its outside is excluded (ionic.py:83-105).
        Must be called before define()."""
        if self.defined:
            raise AssertionError('add_hole_to_phase_field should be called before calling define')
        cols = np.arange(self.width)[np.newaxis, :] - x         # broadcasting instead of a meshgrid: same
        rows = np.arange(self.height)[:, np.newaxis] - y        # float64 values, bit for bit
        dist = np.hypot(cols, rows)
        # smooth indicator of the kept region: outside the disc, or (neg) inside it with a 10x softer edge
        arg = 0.1 * (radius - dist) if neg else dist - radius
        mask = np.array(0.5 * (np.tanh(arg) + 1.0), dtype=np.float32)
        base = np.ones([self.height, self.width], dtype=np.float32) if self.phase is None else self.phase
        # floor at 1e-5: phase_field divides by 4ϕ (ionic.py:104-105)
        self.phase = np.maximum(base * mask, 1e-5)

    # ---- pacing ----------------------------------------------------------------------------------
    def pace_rect(self, loc):
        """rows/cols [r0,r1) x [c0,c1) of a named pacing site, ionic.py:145-160"""
        H, W = self.height, self.width
        table = {
            'left': (0, H, 0, min(5, W)),
            'right': (0, H, max(W - 5, 0), W),
            'top': (0, min(5, H), 0, W),
            'bottom': (max(H - 5, 0), H, 0, W),
            'luq': (1, H // 2, 1, W // 2),
            'llq': (H // 2, H - 1, 1, W // 2),
            'ruq': (1, H // 2, W // 2, W - 1),
            'rlq': (H // 2, H - 1, W // 2, W - 1),
        }
        return table.get(loc)

    def add_pace_op(self, name, loc, v):
        """registers the stimulus `pot = max(pot, s)`, s = v on the site and min_v elsewhere
        (ionic.py:125-163).  Must be called after define()."""
        if not self.defined:
            raise AssertionError('add_hole_to_phase_field should be called after calling define')
        rect = self.pace_rect(loc)
        if rect is None:
            print('undefined pace location')
            rect = (0, 0, 0, 0)                 # s = min_v everywhere, exactly as the reference
        self._ops[name] = ('pace', rect, float(v))

    def fire_op(self, name):
        """runs an operation registered by add_pace_op (ionic.py:165-169)"""
        op = self._ops[name]
        if op[0] == 'pace':
            (r0, r1, c0, c1), v = op[1], op[2]
            self._stepper.pace(r0, r1, c0, c1, v, float(self.min_v))
        else:
            op[1]()

    # ---- main loop -------------------------------------------------------------------------------
    def run(self, im=None, keep_state=False, block=True):
        """generator over ticks, ionic.py:171-245:

            for i in model.run(im):
                if i == s2:
                    model.fire_op('s2')

        One tick = `dt_per_step` sub-steps of dt.  Ticks are enqueued asynchronously; the
        generator only synchronises when the caller reads something back."""
        if not self.defined or self._stepper is None:
            raise AssertionError('run should be called after calling define')
        then = time.time()
        st = self._stepper
        self.samples = int(self.duration / (self.dt_per_step * self.dt))
        every = int(self.dt_per_plot / self.dt_per_step) if im else 0
        watch = {'v0': self.min_v, 'last_spike': 0}
        # run() owns the cadence (ionic.py:199-206) and says so: with a screen the ticks up to the next frame are one series,
        # without one the whole loop is (a loop body that reads something back earlier just ends the series there) — the
        # library launches a declared series at its first tick instead of learning the pattern from the call history
        if not every:
            st.expect(self.samples)
        for i in range(self.samples):
            st.step(1)                           # == sess.run(self.ode_op(i)), ionic.py:203
            yield i
            if every and i % every == 0:         # a frame every dt_per_plot sub-steps, ionic.py:206
                st.expect(min(every, self.samples - 1 - i))
                self._paint(im, i, watch)
        if keep_state:                           # ionic.py:226-229
            self.state = {}
            for s in self._State:
                self.state[s] = self._State[s].eval()
        st.sync()
        elapsed = time.time() - then
        if self.timeline:                        # ionic.py:231-241: trace one more tick, write a Chrome trace
            events = st.trace_tick()             # every launch between two HIP events (+ the halo exchange on row blocks)
            trace = {'traceEvents': [
                {'name': e['name'], 'cat': 'halo' if e.get('host_clock') else 'kernel', 'ph': 'X', 'ts': e['ts'], 'dur': e['dur'],
                 'pid': 0, 'tid': 1 if e.get('host_clock') else 0,
                 'args': {'sub_steps_fused': e['K'], 'tile': '%dx%d' % e['tile'][:2], 'rows_per_wave': e['tile'][2],
                          'ticks': e['ticks'], 'clock': 'host' if e.get('host_clock') else 'HIP events on the stream'}}
                for e in events]}
            with open(self.timeline_name, 'w') as f:
                json.dump(trace, f)
        self.elapsed = elapsed
        print('elapsed: %f sec' % elapsed)
        if block and im:
            im.wait()

    def _paint(self, im, i, watch):
        """one frame + the cycle-length detector at pixel [20, width//2] (ionic.py:207-224)"""
        image = self.image()
        if self.phase is not None:
            image *= self.phase
        im.imshow(image)
        v1 = image[20, self.width // 2]
        if v1 >= 0.5 and watch['v0'] < 0.5:      # upstroke through 0.5 = a wavefront passes
            cl = (i - watch['last_spike']) * self.dt_per_step * self.dt
            if self.cl_observer is None:
                print('wavefront reaches the middle top point at %d, cycle length is %d' % (i, cl))
            else:
                self.cl_observer(i, cl)
            watch['last_spike'] = i
        watch['v0'] = v1

    def millisecond_to_step(self, t):
        """milliseconds -> tick index returned by run(), ionic.py:247-252"""
        return int(t / (self.dt_per_step * self.dt))

    # ---- define(): creates the device-resident state ---------------------------------------------
    def define(self, s1=True):
        """placeholder overridden by the models (ionic.py:254-260)"""
        self.defined = True

    def _flags(self):
        return _lib.FAST if self.fast_math else 0

    def _new_stepper(self, steps_per_tick=0, shard=True):
        """a fibhip handle for the whole grid, or — when a torch.distributed process group with more
        than one rank is initialised — this rank's row block of it (fib_tf_amd/sharded.py)"""
        from .sharded import ShardedStepper, dist_world
        _, world = dist_world()
        if shard and world > 1:
            st = ShardedStepper(self.MODEL_ID, self.height, self.width, self.dt, self.diff, flags=self._flags(),
                                device=self.device, steps_per_tick=steps_per_tick,
                                halo_ticks=getattr(self, 'halo_ticks', 0), library=self._library,
                                halo_mode=getattr(self, 'halo', None))
        else:
            st = _lib.Stepper(self.MODEL_ID, self.height, self.width, self.dt, self.diff, flags=self._flags(),
                              device=self.device, steps_per_tick=steps_per_tick, library=self._library)
        self._configure_stepper(st)
        if self.phase is not None:
            st.set_phase(self.phase)
        return st

    def _configure_stepper(self, st):
        """model-specific constants (BeelerReuter: the Chebyshev table)"""

    def _resume_arrays(self, state):
        """`define(state=...)`: a dict name -> [H, W] array — what `run(keep_state=True)` leaves in `model.state`
        (ionic.py:226-229; court.py:87-89, 623-626 resumes from it) — as the model's arrays in variable order"""
        missing = [n for n in self.VAR_NAMES if n not in state]
        if missing:
            raise KeyError('define(state=...): missing state variables %s' % missing)
        out = []
        for n in self.VAR_NAMES:
            a = np.asarray(state[n], dtype=np.float32)
            if a.shape != (self.height, self.width):
                raise ValueError('define(state=...): %s has shape %s, the grid is %s' % (n, a.shape, (self.height, self.width)))
            out.append(a)
        return out

    def _create(self, init_arrays, steps_per_tick=0):
        """init_arrays: list of [H,W] float32 in the model's variable order"""
        if self._stepper is not None:
            self._stepper.close()
        st = self._new_stepper(steps_per_tick)
        st.set_state(-1, np.stack([np.asarray(a, np.float32) for a in init_arrays]))
        self._stepper = st
        self.dt_per_step = st.steps_per_tick
        self._State = {n: StateVar(self, i, n) for i, n in enumerate(self.VAR_NAMES)}
        return st

    def image(self):
        pass

    def pot(self):
        pass

    def ode_op(self, tick):
        """kept for API symmetry (ionic.py:277-286): the tick operation is the fibhip handle"""
        return self._stepper

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def jit_scope(self):
        """the reference returns an XLA scope when available (ionic.py:294-307); fusion is what
        the HIP kernel does already, so this is the no-op context"""
        return self
