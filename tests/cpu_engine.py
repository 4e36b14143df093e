"""TEST INFRASTRUCTURE: a CPU stand-in for the HIP engine so that the HOST logic of
fib_tf_amd/sharded.py (row slicing, halo exchange, gather, pacing in global coordinates) can be
exercised with gloo on CPU tensors.  Arithmetic comes from the oracle.  It lives under tests/
because the product must never route through the oracle.

The oracle treats the edges of the array it is given as the domain boundary (row 0 is overwritten
by row 1 and mirrored).  The engine therefore pads a ghost edge with one throw-away row: the
boundary treatment then spoils that row, and the error it feeds inward travels one row per sub-step,
so after g sub-steps it has consumed exactly the g ghost rows and the owned rows are exact — the
same argument the temporally blocked HIP kernel relies on.
"""
import numpy as np
import torch

import oracle

NVAR = {0: 4, 1: 8, 2: 21}
STEPS = {0: 10, 1: 5, 2: 1}
CHEBY, SKIP, CHRONIC = 1, 2, 4


class OracleEngine:
    @staticmethod
    def nvar(model_id):
        return NVAR[model_id]

    @staticmethod
    def default_steps(model_id):
        return STEPS[model_id]

    def __init__(self, model_id, height, width, dt, diff, flags, steps_per_tick, global_height, row_offset,
                 ghost_top, ghost_bottom, device):
        self.model, self.h, self.w, self.dt, self.diff, self.flags = model_id, height, width, dt, diff, flags
        self.spt, self.Hg, self.lo, self.gt, self.gb = steps_per_tick, global_height, row_offset, ghost_top, ghost_bottom
        n = NVAR[model_id]
        self.slabs = [torch.zeros((n, height, width), dtype=torch.float32) for _ in range(2)]
        self.cur, self.open = 0, False
        self.phi, self.cheb = None, None
        ghosts = [x for x in (ghost_top, ghost_bottom) if x]
        self.cycle = (min(ghosts) // steps_per_tick) if ghosts else 1
        self.cpos = 0

    # -- Stepper surface -----------------------------------------------------------------------
    def set_phase(self, phi):
        self.phi = None if phi is None else np.ascontiguousarray(phi, np.float32)

    def set_state(self, var, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, np.float32))
        for s in self.slabs:
            if var < 0:
                s.copy_(t)
            else:
                s[var].copy_(t)

    def get_state(self, var=-1):
        s = self.slabs[self.cur]
        return (s if var < 0 else s[var]).numpy().copy()

    def set_consts(self, tbl):
        self.cheb = np.ascontiguousarray(tbl, np.float32).reshape(12, 9)

    def halo_vars(self):
        return NVAR[self.model] if (self.spt > 1 or self.cycle > 1) else 1

    def halo_due(self):
        return bool(self.gt or self.gb) and self.cpos == self.cycle - 1

    def launch_plan(self):
        return self.spt, 1

    def _pad(self, a):
        parts = ([a[..., :1, :]] if self.gt else []) + [a] + ([a[..., -1:, :]] if self.gb else [])
        return np.ascontiguousarray(np.concatenate(parts, axis=-2))

    def _unpad(self, a):
        return a[..., (1 if self.gt else 0):a.shape[-2] - (1 if self.gb else 0), :]

    def step_edges(self):
        a = self._pad(self.slabs[self.cur].numpy())
        phi_keep, self.phi = self.phi, (None if self.phi is None else self._pad(self.phi))
        try:
            a = self._advance(a)
        finally:
            self.phi = phi_keep
        self.slabs[self.cur ^ 1].copy_(torch.from_numpy(np.ascontiguousarray(self._unpad(a))))
        self.open = True

    def _advance(self, a):
        if self.model == 0:
            assert self.spt % 1 == 0
            oracle.fenton_run(a, self.dt, self.diff, self.phi, self.spt)
        elif self.model == 1:
            assert self.spt == 5
            oracle.br_run(a, self.dt, self.diff, self.phi, self.cheb if self.flags & CHEBY else None,
                          bool(self.flags & SKIP), 1)
        else:
            new = oracle.court_step(a, self.dt, self.diff, self.phi, bool(self.flags & CHRONIC))
            a[oracle.COURT_FAST] = new[oracle.COURT_FAST]
        return a

    def step_interior(self):
        pass

    def step(self, n=1):
        for _ in range(n):
            self.step_edges()
            self.step_interior()
            self.step_commit()

    def step_commit(self):
        self.cur ^= 1
        self.cpos = (self.cpos + 1) % self.cycle
        self.open = False

    def step_slow(self):
        a = self.slabs[self.cur].numpy()
        new = self._unpad(oracle.court_step(self._pad(a), self.dt, self.diff,
                                            None if self.phi is None else self._pad(self.phi),
                                            bool(self.flags & CHRONIC)))
        slow = [i for i in range(21) if i not in oracle.COURT_FAST]
        a[slow] = new[slow]

    def next_buf(self, var):
        return self.cur ^ 1, None

    def state_buf(self, var):
        return self.cur, None

    def pace(self, r0, r1, c0, c1, v, min_v):
        a = self.slabs[self.cur][0].numpy()
        a[:] = oracle.pace(a, r0 - self.lo, r1 - self.lo, c0, c1, v, min_v)

    def probe(self, var, row, col):
        return np.float32(self.slabs[self.cur][var, row, col].item())

    # -- HipEngine surface ---------------------------------------------------------------------
    interleaved = False

    def var_view(self, idx, v):
        return self.slabs[idx][v]

    def to_host(self, t):
        return t.numpy()

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32))

    def empty(self, shape):
        return torch.empty(shape, dtype=torch.float32)

    def device_sync(self):
        pass

    def stream_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def sync(self):
        pass

    def close(self):
        pass
