#!/usr/bin/env python3
"""GPU box: host and device cost of ONE exchange tick (kernel + halo exchange + commit) for the two halo paths, on a
one-rank RCCL group whose neighbours on both sides are the rank itself (the only RCCL exchange a one-GPU box can run):
torch.distributed.batch_isend_irecv on the slab views (default) vs the library's own grouped ncclSend/ncclRecv on the
compute stream (FIBTF_HALO=direct).  Weak-scaling bench block: 512 owned rows x 512 columns, 10-row ghost zones
exchanged every tick (halo_ticks = 1, so that every tick is an exchange tick)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from fib_tf_amd import _lib  # noqa: E402
from fib_tf_amd.sharded import HipEngine, init_from_env  # noqa: E402

init_from_env()
for W, rows, g in ((512, 512, 10), (512, 512, 40), (4096, 512, 40)):      # the last one: a rank's block of BASELINE configs[3] on 8 GPUs
    H = rows + 2 * g
    eng = HipEngine(_lib.FENTON4V, H, W, 0.1, 1.5, _lib.FAST, 10, 4 * rows, rows, g, g, 0)
    rng = np.random.default_rng(0)
    eng.set_state(-1, rng.uniform(0, 1, (4, H, W)).astype(np.float32))
    eng.set_phase(rng.uniform(0.5, 1, (H, W)).astype(np.float32))
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    eng.st.comm_init(eng.st.comm_unique_id(path), 0, 1, path)
    b = H - g
    cycle = g // 10

    def tick(direct):
        for _ in range(cycle - 1):
            eng.step(1)                                    # mid-cycle ticks: no exchange
        eng.step_edges()
        if direct == 2:                                    # torch transport, the interior launched BEFORE the messages are posted
            slab = eng.slabs[eng.next_buf(0)[0]]
            eng.step_interior()
            ops = [dist.P2POp(dist.isend, slab[g:2 * g], 0), dist.P2POp(dist.irecv, slab[:g], 0),
                   dist.P2POp(dist.isend, slab[b - g:b], 0), dist.P2POp(dist.irecv, slab[b:], 0)]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        elif direct:
            eng.comm_exchange(0, 0)
            eng.step_interior()
        else:
            slab = eng.slabs[eng.next_buf(0)[0]]
            ops = [dist.P2POp(dist.isend, slab[g:2 * g], 0), dist.P2POp(dist.irecv, slab[:g], 0),
                   dist.P2POp(dist.isend, slab[b - g:b], 0), dist.P2POp(dist.irecv, slab[b:], 0)]
            reqs = dist.batch_isend_irecv(ops)
            eng.step_interior()
            for r in reqs:
                r.wait()
        eng.step_commit()

    with eng.stream_ctx():
        for direct in (False, 2, True):
            for _ in range(20):
                tick(direct)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                tick(direct)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print('%dx%d block, ghost %2d rows (%d-tick cycle), %-28s host %6.1f us per cycle (enqueue), %6.1f us per cycle drained = %5.1f us per tick'
                  % (rows, W, g, cycle, {False: 'torch batch_isend_irecv:', 2: 'torch, interior first:', True: 'library ncclSend/ncclRecv:'}[direct], (t1 - t0) / n * 1e6,
                     (t2 - t0) / n * 1e6, (t2 - t0) / n / cycle * 1e6), flush=True)
    eng.st.close()
dist.destroy_process_group()
