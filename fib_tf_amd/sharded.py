"""Row-block domain decomposition across the GPUs of one node: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI), one fibhip handle per process.

Decomposition.  Rank r owns a contiguous block of rows of every state array and keeps
`g = halo_ticks * steps_per_tick` ghost rows of its upper and lower neighbour.  The kernels already
recompute a rim redundantly (temporal blocking, kernels.hpp); the ghost zone extends that idea across
ticks: tick j of a cycle also advances the (halo_ticks-1-j) * steps_per_tick ghost rows next to the
block, so the neighbours' rows are needed only every `halo_ticks` ticks.  Then each rank sends the g
outermost owned rows of ALL state arrays and receives the neighbour's into its ghost rows — one
point-to-point message pair per neighbour, no collective on the data path.  A 40-row message costs
far less than four 10-row ones: at these sizes an exchange is latency, not bandwidth.

Overlap.  `step_edges` computes the strips the neighbours wait for on the main stream; the sends
are posted right behind them; `step_interior` runs the bulk of the block on a second HIP stream
while the messages are on the wire; `step_commit` joins the two.

PyTorch is plumbing here: it owns the two device slabs (so RCCL can address them), the stream and
the process group.  The arithmetic is libfibhip's.  The `engine_factory` hook exists so that the
host logic of this file (slicing, exchange, gather) can be exercised on CPU tensors with gloo; the
product default is the HIP engine and nothing else.
"""
import os
import time

import numpy as np

from . import _lib


def dist_world():
    """(rank, world) of the initialised default process group, or (0, 1)"""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def row_blocks(height, world):
    """[(first_row, n_rows)] per rank: as even as possible, earlier ranks take the remainder"""
    base, rem = divmod(height, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((r0, n))
        r0 += n
    return out


class HipEngine:
    """the product engine: a fibhip handle on torch-owned device slabs and torch's current stream"""

    def __init__(self, model_id, height, width, dt, diff, flags, steps_per_tick, global_height, row_offset,
                 ghost_top, ghost_bottom, device, library=None):
        import torch
        self.torch = torch
        self.dev = torch.device('cuda', device)
        L = library or _lib.lib()
        nvar = _lib.check(L.fibhip_nvar(model_id), L)
        # row-interleaved slab [rows][nvar][width] (FIBHIP_ROW_INTERLEAVED): the g halo rows of ALL arrays
        # are one contiguous block, i.e. one RCCL message per neighbour and direction
        self.interleaved = True
        self.slabs = [torch.zeros((height, nvar, width), dtype=torch.float32, device=self.dev) for _ in range(2)]
        # a stream of our own (torch's default stream has the null handle, which fibhip reads as
        # "create one"): kernels, halo packing and RCCL's stream dependencies all hang off this one
        self.stream = torch.cuda.Stream(self.dev)
        torch.cuda.synchronize(self.dev)                    # the zero-fill above ran on the default stream
        stream = self.stream.cuda_stream
        self.st = _lib.Stepper(model_id, height, width, dt, diff, flags=flags | _lib.ROW_INTERLEAVED, device=device,
                               steps_per_tick=steps_per_tick, global_height=global_height, row_offset=row_offset,
                               ghost_top=ghost_top, ghost_bottom=ghost_bottom, stream=stream,
                               ext_slabs=(self.slabs[0].data_ptr(), self.slabs[1].data_ptr()), library=library)
        self.nvar = nvar

    # thin forwards
    def __getattr__(self, name):
        return getattr(self.st, name)

    def var_view(self, idx, v):
        """[rows, width] view of state array v in slab idx"""
        return self.slabs[idx][:, v, :]

    def to_host(self, t):
        return t.detach().cpu().numpy()

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev)

    def empty(self, shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.dev)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def device_sync(self):
        self.st.sync()
        self.torch.cuda.synchronize(self.dev)


class ShardedStepper:
    """same surface as `_lib.Stepper`, for one row block of a grid shared by all ranks"""

    def __init__(self, model_id, height, width, dt, diff, flags=0, device=0, steps_per_tick=0,
                 engine_factory=None, group=None, halo_ticks=0, library=None, halo_mode=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.H, self.width = height, width
        self.height = height                               # global, like Stepper.height for one device
        L = None
        if engine_factory is None:
            L = library or _lib.lib()
            nvar = _lib.check(L.fibhip_nvar(model_id), L)
            spt = steps_per_tick or _lib.check(L.fibhip_default_steps_per_tick(model_id), L)
        else:
            nvar, spt = engine_factory.nvar(model_id), steps_per_tick or engine_factory.default_steps(model_id)
        self.nvar, self.steps_per_tick = nvar, spt
        # Two halo schemes.  'ghost' (default): a ghost zone several ticks deep, ALL arrays exchanged once per cycle, the
        # ticks in between recompute the ghost rows they still need.  'rows1': the scheme BASELINE's north_star spells
        # out — ONE ghost row of the potential alone, exchanged after every SUB-step (the other arrays are pointwise:
        # nobody reads their ghost rows), one sub-step per launch, the block's edge rows first and its interior on the
        # second stream while the rows travel.  Same arithmetic, bit-identical results; kept so that the two can be
        # timed against each other on a node with xGMI.
        self.halo_mode = halo_mode or os.environ.get('FIBTF_HALO_MODE', 'ghost')
        if self.halo_mode not in ('ghost', 'rows1'):
            raise ValueError("halo mode %r: 'ghost' or 'rows1'" % (self.halo_mode,))
        self.sub = 1                                       # engine ticks per tick of the model
        if self.halo_mode == 'rows1':
            if flags & _lib.SKIP:
                raise ValueError("'rows1' runs one sub-step per engine tick: Beeler-Reuter's `skip` schedule (slow gates on "
                                 "the first sub-step of a tick) needs the tick as a unit — use the ghost-zone scheme")
            self.sub, spt, halo_ticks = spt, 1, 1
        self.blocks = row_blocks(height, self.world)
        self.row0, self.rows = self.blocks[self.rank]
        rows_min = min(n for _, n in self.blocks)
        if rows_min < max(spt, 2):
            raise ValueError('row blocks of %d rows are thinner than the %d-row halo: use fewer ranks or a taller '
                             'grid' % (rows_min, spt))
        # ghost zone = m ticks deep: the halo is exchanged every m-th tick only, the ticks in between recompute
        # the ghost rows they still need (a message costs far more than a few extra rows of compute).  Tick j of a cycle
        # advances (m-1-j)*spt extra rows per neighbour, on average spt*(m-1)/2: unless the caller fixes m, it is the
        # deepest zone (at most 4 ticks) whose extra rows stay within half of the block's own rows, 2 * spt*(m-1)/2 <=
        # rows/2 (512 rows over 8 ranks: 64-row blocks, m = 4: 30 extra rows per tick on average).
        if halo_ticks and int(halo_ticks) > 0:
            m = max(1, min(int(halo_ticks), rows_min // spt))
        else:
            m = max(1, min(4, rows_min // spt, 1 + rows_min // (2 * spt)))
        self.halo_ticks = m
        self.eng_spt = spt                                  # sub-steps per ENGINE tick (1 under 'rows1')
        self.g = m * spt
        self.gt = self.g if self.rank > 0 else 0
        self.gb = self.g if self.rank < self.world - 1 else 0
        self.lo = self.row0 - self.gt                       # global row of local row 0
        self.lh = self.rows + self.gt + self.gb             # local slab height
        if engine_factory is None:
            self.eng = HipEngine(model_id, self.lh, width, dt, diff, flags, spt, height, self.lo, self.gt, self.gb, device,
                                 library=library)
        else:
            self.eng = engine_factory(model_id, self.lh, width, dt, diff, flags, spt, height, self.lo, self.gt, self.gb,
                                      device)
        self.halo_n = self.eng.halo_vars()
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank < self.world - 1 else None
        # RCCL moves device buffers directly.  A backend without device point-to-point (gloo: used to
        # rehearse this driver with several ranks on ONE GPU, and by the CPU tests) gets host staging.
        self.staged = dist.get_backend(group) != 'nccl'
        shape = (self.halo_n, self.g, width)
        mk = (lambda: torch.empty(shape, dtype=torch.float32)) if self.staged else (lambda: self.eng.empty(shape))
        self.recv_up = mk() if self.up is not None else None
        self.recv_down = mk() if self.down is not None else None
        self._op_cache = {}
        self._trace_t0, self._trace_ev = None, []
        self.comm_s = 0.0
        # Halo transport on RCCL.  Default: torch.distributed's batch_isend_irecv on the slab views (one contiguous
        # [g, nvar, width] block each way).  FIBTF_HALO=direct (opt-in): the library issues the grouped
        # ncclSend/ncclRecv itself on the compute stream (include/fibhip.h fibhip_comm_*): one C call and one RCCL
        # kernel per exchange, 92 us per 4-tick cycle against 133 us through torch on one GPU talking to itself
        # (tools/exchange_tick_cost.py) — opt-in because two real ranks have not exchanged through it on hardware
        # yet.  Any rank failing to set the direct path up sends ALL ranks back to the torch path, collectively.
        self.rccl_direct = False
        if (not self.staged and engine_factory is None and os.environ.get('FIBTF_HALO', 'torch') == 'direct'
                and (self.world > 1 or os.environ.get('FIBTF_HALO_SELFTEST') == '1')):    # (self-test: one-rank group)
            self.rccl_direct = self._setup_direct(group)
        self.halo_path = 'library ncclSend/ncclRecv on the compute stream' if self.rccl_direct else (
            'host-staged (no device point-to-point on this backend)' if self.staged else 'torch.distributed batch_isend_irecv')

    def _agree(self, ok, group):
        """True iff every rank says ok"""
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.eng.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    def _setup_direct(self, group):
        """communicator for the library-issued exchange.  Everything that can fail on ONE rank (dlopen/dlsym of
        librccl, the handle's flags, ghost rows vs neighbours) is checked locally and agreed on BEFORE any rank
        enters the collective ncclCommInitRank: a rank that raised alone would leave the others blocked in it."""
        torch, dist = self.torch, self.dist
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        uid, ok = None, True
        try:                                                # bind RCCL; local checks; rank 0 draws the communicator id
            self.eng.st.comm_open(path)
            self.eng.st.comm_check(self.rank, self.world)
            if self.rank == 0:
                uid = self.eng.st.comm_unique_id(path)
        except Exception as e:                              # noqa: BLE001  (any failure = use the other transport)
            ok = False
            print('fib_tf_amd.sharded: direct RCCL path unavailable on rank %d (%s)' % (self.rank, e))
        if not self._agree(ok, group):
            return False
        box = [uid if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        try:                                                # collective from here on: only RCCL itself can fail now
            self.eng.st.comm_init(box[0], self.rank, self.world, path)
        except Exception as e:                              # noqa: BLE001
            ok = False
            print('fib_tf_amd.sharded: communicator setup failed on rank %d (%s)' % (self.rank, e))
        if not self._agree(ok, group):
            self.eng.st.comm_free()
            return False
        return True

    # ---- data movement between the global arrays and this block ---------------------------------
    def _local(self, arr):
        return np.ascontiguousarray(arr[..., self.lo:self.lo + self.lh, :], dtype=np.float32)

    def set_phase(self, phi):
        self.eng.set_phase(None if phi is None else self._local(np.asarray(phi)))

    def set_state(self, var, arr):
        self.eng.set_state(var, self._local(np.asarray(arr)))

    def get_state(self, var=-1):
        """the WHOLE array(s), gathered on every rank (image()/eval() are global in the reference)"""
        torch, dist = self.torch, self.dist
        local = self.eng.get_state(var)                                    # numpy, local rows
        own = local[..., self.gt:self.gt + self.rows, :]
        maxr = max(n for _, n in self.blocks)
        lead = own.shape[:-2]
        pad = np.zeros(lead + (maxr, self.width), np.float32)
        pad[..., :self.rows, :] = own
        mine = torch.from_numpy(pad) if self.staged else self.eng.from_host(pad)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        out = np.empty(lead + (self.H, self.width), np.float32)
        for (r0, n), p in zip(self.blocks, parts):
            out[..., r0:r0 + n, :] = p.cpu().numpy()[..., :n, :]
        return out

    def set_consts(self, tbl):
        self.eng.set_consts(tbl)

    # ---- one tick ----------------------------------------------------------------------------------
    def _slab_view(self, var, nxt):
        idx, _ = self.eng.next_buf(var) if nxt else self.eng.state_buf(var)
        return self.eng.var_view(idx, var)

    def _tick(self):
        if not self.eng.halo_due():                         # mid-cycle tick: one C call, no torch involved
            self.eng.step(1)
            return
        with self.eng.stream_ctx():
            self._tick_on_stream()

    def _p2p_ops(self, idx):
        """RCCL fast path, cached per slab index: one contiguous [g, nvar, width] block each way"""
        key = idx
        if key not in self._op_cache:
            dist, g, b = self.dist, self.g, self.gt + self.rows
            slab, ops = self.eng.slabs[idx], []
            if self.up is not None:
                ops += [dist.P2POp(dist.isend, slab[self.gt:self.gt + g], self.up, self.group),
                        dist.P2POp(dist.irecv, slab[:g], self.up, self.group)]
            if self.down is not None:
                ops += [dist.P2POp(dist.isend, slab[b - g:b], self.down, self.group),
                        dist.P2POp(dist.irecv, slab[b:], self.down, self.group)]
            self._op_cache[key] = ops
        return self._op_cache[key]

    def _tick_on_stream(self):
        torch, dist, e, g = self.torch, self.dist, self.eng, self.g
        e.step_edges()
        t0 = time.perf_counter()
        b = self.gt + self.rows
        idxs = {e.next_buf(v)[0] for v in range(self.halo_n)}
        one_block = getattr(e, 'interleaved', False) and self.halo_n == self.nvar and len(idxs) == 1
        direct = not self.staged and one_block
        if self.staged and one_block and os.environ.get('FIBTF_HALO') == 'plan':
            # test mode: the library's own message description (fibhip_halo_plan — what fibhip_comm_exchange hands
            # to RCCL) executed through host staging, so that several ranks on ONE GPU verify its row arithmetic
            msgs, idx = e.halo_plan(self.up, self.down)
            self.plan_exchanges = getattr(self, 'plan_exchanges', 0) + 1
            flat = e.slabs[idx].view(-1)
            ops, recvs, keep = [], [], []
            for off, cnt, peer, send in msgs:
                if send:
                    t = flat[off:off + cnt].cpu()
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, peer, self.group))
                else:
                    t = torch.empty(cnt, dtype=torch.float32)
                    recvs.append((off, cnt, t))
                    ops.append(dist.P2POp(dist.irecv, t, peer, self.group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            e.step_interior()
            for r in reqs:
                r.wait()
            for off, cnt, t in recvs:
                flat[off:off + cnt].copy_(t)
        elif direct and self.rccl_direct:
            e.comm_exchange(self.up, self.down)             # stream-ordered after the kernels of step_edges
            e.step_interior()
        elif direct:
            ops = self._p2p_ops(next(iter(idxs)))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            e.step_interior()                               # overlaps with the messages
            for r in reqs:
                r.wait()
        else:
            # general path: pack the halo rows of the exchanged arrays (they may sit in different slabs when
            # sub-steps are launched one by one), optionally stage through the host (non-RCCL backends)
            def dev(t):
                return t.cpu() if self.staged else t
            ops, keep = [], []
            if self.up is not None:
                s = dev(torch.stack([self._slab_view(v, True)[self.gt:self.gt + g] for v in range(self.halo_n)]))
                ops += [dist.P2POp(dist.isend, s, self.up, self.group),
                        dist.P2POp(dist.irecv, self.recv_up, self.up, self.group)]
                keep.append(s)
            if self.down is not None:
                s = dev(torch.stack([self._slab_view(v, True)[b - g:b] for v in range(self.halo_n)]))
                ops += [dist.P2POp(dist.isend, s, self.down, self.group),
                        dist.P2POp(dist.irecv, self.recv_down, self.down, self.group)]
                keep.append(s)
            reqs = dist.batch_isend_irecv(ops) if ops else []
            e.step_interior()
            for r in reqs:
                r.wait()
            for v in range(self.halo_n):
                if self.up is not None:
                    self._slab_view(v, True)[:g].copy_(self.recv_up[v])
                if self.down is not None:
                    self._slab_view(v, True)[b:].copy_(self.recv_down[v])
        self.comm_s += time.perf_counter() - t0
        if self._trace_t0 is not None:                      # trace_tick: the exchange as an event of its own (host clock)
            self._trace_ev.append({'name': 'halo exchange: %s, %d rows x %d arrays x %d columns per neighbour' % (
                self.halo_path, g, self.halo_n, self.width), 'ts': (t0 - self._trace_t0) * 1e6,
                'dur': (time.perf_counter() - t0) * 1e6, 'K': 0, 'tile': (0, 0, 0), 'ticks': 0, 'host_clock': True})
        e.step_commit()

    def step(self, nticks=1):
        for _ in range(nticks * self.sub):
            self._tick()

    def trace_tick(self):
        """one whole exchange cycle (halo_ticks ticks): this rank's launches between HIP events + the halo exchange as its
        own event, timed on the host clock from its first enqueue to the last completed message"""
        if not hasattr(self.eng, 'trace_begin'):
            self.step(max(1, self.halo_ticks // self.sub))
            return []
        self._trace_ev = []
        self.eng.trace_begin()
        self._trace_t0 = time.perf_counter()
        try:
            self.step(self.halo_ticks)                      # ('rows1': one tick = its sub-steps, each with an exchange)
        finally:
            self._trace_t0 = None
        return self.eng.trace_end() + self._trace_ev

    def step_slow(self):
        self.eng.step_slow()                                # pointwise: no halo needed

    def step_mode(self, mode):
        self.eng.step_mode(mode)                            # pointwise as well

    def pace(self, r0, r1, c0, c1, v, min_v):
        self.eng.pace(r0, r1, c0, c1, v, min_v)             # global rectangle; the kernel offsets rows

    def probe(self, var, row, col):
        owner = next(i for i, (r0, n) in enumerate(self.blocks) if r0 <= row < r0 + n)
        val = self.torch.zeros((1,), dtype=self.torch.float32) if self.staged else self.eng.empty((1,))
        if owner == self.rank:
            val.fill_(float(self.eng.probe(var, row - self.lo, col)))
        self.dist.broadcast(val, owner, group=self.group)
        return np.float32(val.cpu().numpy()[0])

    def sync(self):
        self.eng.device_sync()

    def expect(self, nticks):
        """(row blocks advance tick by tick, exchange in between: nothing to declare)"""

    def time_steps(self, nticks):
        self.sync()
        self.dist.barrier(group=self.group)
        l0 = self.launches()
        t0 = time.perf_counter()
        self.step(nticks)
        self.sync()
        return (time.perf_counter() - t0) * 1e3, self.launches() - l0

    def launches(self):
        return getattr(self.eng, 'launch_count', lambda: 0)()

    def launch_plan(self):
        return self.eng.launch_plan()

    def close(self):
        if self.rccl_direct:
            self.eng.st.comm_free()
            self.rccl_direct = False
        if hasattr(self.eng, 'close'):
            self.eng.close()


def init_from_env():
    """process group + device for a rank started by `python -m torch.distributed.run`"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # FIBTF_DIST_BACKEND=gloo + FIBTF_ONE_DEVICE=1 rehearse this path with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); production is always nccl = RCCL, one GPU per rank
    backend = os.environ.get('FIBTF_DIST_BACKEND', 'nccl')
    if os.environ.get('FIBTF_ONE_DEVICE') == '1':
        local = 0
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local
