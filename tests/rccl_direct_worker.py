"""worker for tests/test_gpu_sharded.py::test_direct_rccl_exchange_self: the library's own ncclSend/ncclRecv halo path
(include/fibhip.h fibhip_comm_*) with a one-rank communicator whose neighbours on both sides are the rank itself —
the mechanics a one-GPU box can run: dlopen of the process's librccl, unique id, communicator, one grouped exchange on
the handle's stream between step_edges and step_commit, data in place."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out):
    import torch
    from fib_tf_amd import _lib
    from fib_tf_amd.sharded import HipEngine
    torch.cuda.set_device(0)
    H, W, g = 96, 64, 20
    eng = HipEngine(_lib.FENTON4V, H, W, 0.1, 1.5, _lib.FAST, 10, 4 * H, H, g, g, 0)
    rng = np.random.default_rng(3)
    eng.set_state(-1, rng.uniform(0, 1, (4, H, W)).astype(np.float32))
    eng.set_phase(rng.uniform(0.5, 1, (H, W)).astype(np.float32))
    path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    uid = eng.st.comm_unique_id(path)
    eng.st.comm_init(uid, 0, 1, path)
    with eng.stream_ctx():
        eng.step(1)
        assert eng.halo_due()
        eng.step_edges()
        idx = {eng.next_buf(v)[0] for v in range(eng.halo_vars())}
        slab = eng.slabs[next(iter(idx))]
        torch.cuda.synchronize()
        before = slab.clone()
        eng.comm_exchange(0, 0)                            # both neighbours = myself: sends pair with recvs in order
        eng.step_interior()
        eng.step_commit()
        torch.cuda.synchronize()
        b = H - g
        ok = bool(torch.equal(slab[:g], before[g:2 * g]) and torch.equal(slab[b:], before[b - g:b])
                  and torch.equal(slab[g:b], before[g:b]))
        eng.step(2)
        torch.cuda.synchronize()
    # wrong usage is refused, not executed: outside an open tick; neighbours that do not match the ghost rows
    refused = 0
    try:
        eng.comm_exchange(0, 0)
    except _lib.FibhipError:
        refused += 1
    with eng.stream_ctx():
        eng.step(1)                                        # mid-cycle tick, then the exchange tick of the cycle
        eng.step_edges()
        try:
            eng.comm_exchange(None, 0)
        except _lib.FibhipError:
            refused += 1
        eng.comm_exchange(0, 0)
        eng.step_interior()
        eng.step_commit()
        torch.cuda.synchronize()
    np.save(out, np.array([ok, refused == 2]))


if __name__ == '__main__':
    main(sys.argv[1])
