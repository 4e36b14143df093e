"""worker for tests/test_gpu_sharded.py::test_rccl_self_exchange_on_slab_views: ONE rank on the RCCL backend sends the
halo rows of a HipEngine slab to itself with exactly the P2POp pattern of ShardedStepper._p2p_ops (contiguous row
ranges of the row-interleaved slab, received in place, on the engine's own stream) — the part of the multi-GPU path a
one-GPU box can execute for real."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import torch
    import torch.distributed as dist
    from fib_tf_amd import _lib
    from fib_tf_amd.sharded import HipEngine, init_from_env
    init_from_env()                                        # backend nccl (= RCCL), device 0
    assert dist.get_backend() == 'nccl'
    H, W, g = 96, 64, 20
    # a middle row block of a taller grid: ghost rows on both sides, as on an interior rank
    eng = HipEngine(_lib.FENTON4V, H, W, 0.1, 1.5, _lib.FAST, 10, 4 * H, H, g, g, 0)
    rng = np.random.default_rng(3)
    init = rng.uniform(0, 1, (4, H, W)).astype(np.float32)
    eng.set_state(-1, init)
    eng.set_phase(rng.uniform(0.5, 1, (H, W)).astype(np.float32))
    with eng.stream_ctx():
        eng.step_edges()                                   # halo_due: cycle of 2 ticks -> not due on the first
        eng.step_interior()
        eng.step_commit()
        assert eng.halo_due()
        eng.step_edges()
        idx = {eng.next_buf(v)[0] for v in range(eng.halo_vars())}
        assert len(idx) == 1
        slab = eng.slabs[next(iter(idx))]
        b = H - g
        before = slab.clone()
        # "up" neighbour = myself: my top owned rows land in my bottom ghost rows and vice versa
        ops = [dist.P2POp(dist.isend, slab[g:2 * g], 0), dist.P2POp(dist.irecv, slab[b:], 0),
               dist.P2POp(dist.isend, slab[b - g:b], 0), dist.P2POp(dist.irecv, slab[:g], 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        eng.step_interior()
        eng.step_commit()
        torch.cuda.synchronize()
        ok = bool(torch.equal(slab[b:], before[g:2 * g]) and torch.equal(slab[:g], before[b - g:b])
                  and torch.equal(slab[g:b], before[g:b]))
        eng.step(1)                                        # and the engine keeps stepping on the exchanged rows
        torch.cuda.synchronize()
    # the production set-up of the library's own transport (ShardedStepper._setup_direct): agreement all-reduce,
    # id broadcast, communicator — on a one-rank group, where there is nothing to exchange afterwards
    from fib_tf_amd.sharded import ShardedStepper
    os.environ['FIBTF_HALO_SELFTEST'] = '1'
    default = ShardedStepper(_lib.FENTON4V, 64, 48, 0.1, 1.5, flags=_lib.FAST, device=0)
    default_ok = (not default.rccl_direct) and 'batch_isend_irecv' in default.halo_path     # the default transport
    default.close()
    os.environ['FIBTF_HALO'] = 'direct'                    # the library-issued exchange is opt-in
    ss = ShardedStepper(_lib.FENTON4V, 64, 48, 0.1, 1.5, flags=_lib.FAST, device=0)
    ss.set_state(-1, rng.uniform(0, 1, (4, 64, 48)).astype(np.float32))
    ss.step(3)
    full = ss.get_state(-1)
    setup_ok = bool(default_ok and ss.rccl_direct and 'ncclSend' in ss.halo_path and np.isfinite(full).all())
    np.save(out, np.array([ok, slab.is_contiguous(), slab[g:2 * g].is_contiguous(), setup_ok]))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(int(sys.argv[1]), sys.argv[2])
