#!/usr/bin/env python3
"""GPU box: host-side and device-side cost of one torch.distributed batch_isend_irecv round trip on the RCCL
backend (one rank sending to itself — the only RCCL point-to-point a one-GPU box can run), for the halo message
size of the weak-scaling bench (40 rows x 4 arrays x 512 columns, float32)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
n = 40 * 4 * 512
a, b = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
c, d = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    for _ in range(5):
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(N):
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    e1.record(st)
    torch.cuda.synchronize()
print('batch_isend_irecv(2 sends + 2 recvs of %d KiB) + wait: host %.1f us per exchange (enqueue only), %.1f us drained; '
      'device-side %.1f us per exchange' % (n * 4 // 1024, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, e0.elapsed_time(e1) / N * 1e3))
dist.destroy_process_group()
