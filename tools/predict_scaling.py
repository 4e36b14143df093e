#!/usr/bin/env python3
"""GPU box (one MI355X): a PREDICTED 1/2/4/8-GPU table for `bench.py --gpus N`, to be confronted with the first
SCALE record of an 8-GPU node.  Measured here: the kernels of ONE interior rank's row block (owned + ghost rows, row
offset, interleaved slab — exactly the handle a rank runs) as a stand-alone handle, HIP-event timed, for both halo schemes:
  ghost   multi-tick ghost zone of all arrays, halo depth chosen from the block height (sharded.py's rule)
  rows1   one ghost row of the potential, one sub-step per launch, an exchange after every sub-step (north_star's scheme)
Not measurable on one GPU: the exchange between two devices.  Its cost is MODELLED from what one device talking to itself
costs (profiles/r01_exchange_tick_cost.txt: torch batch_isend_irecv 75 us latency per exchange, the library's own grouped
ncclSend/ncclRecv 30 us) plus message bytes over one xGMI link (assumed 50 GB/s sustained per direction of the 153 GB/s
peak).  Tall blocks overlap the exchange with the interior (second stream): tick = max(kernels, exchange + edge strips);
short blocks (one launch) add it: tick = kernels + exchange / halo_ticks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fib_tf_amd import _lib

XGMI_GBS = 50.0
LAT_US = {'torch': 75.0, 'library': 30.0}
SPT = 10


def block_time(size, world, scheme):
    rows = size // world
    if scheme == 'rows1':
        spt, m = 1, 1
    else:
        spt = SPT
        m = max(1, min(4, rows // spt, 1 + rows // (2 * spt)))
    g = m * spt if world > 1 else 0
    interior = world > 2
    gt = g if world > 1 else 0
    gb = g if interior or world == 2 and False else (g if world > 1 and interior else 0)
    if world == 2:
        gt, gb = g, 0                      # rank 1 of 2: one neighbour
    H = rows + gt + gb
    flags = _lib.FAST | (_lib.ROW_INTERLEAVED if world > 1 else 0)
    st = _lib.Stepper(_lib.FENTON4V, H, size, 0.1, 1.5, flags=flags, steps_per_tick=spt, global_height=size,
                      row_offset=(rows - gt) if world > 1 else 0, ghost_top=gt, ghost_bottom=gb)
    rng = np.random.default_rng(0)
    st.set_state(-1, rng.uniform(0, 1, (4, H, size)).astype(np.float32))
    st.set_phase(rng.uniform(0.5, 1, (H, size)).astype(np.float32))
    cyc = m
    st.step(4 * cyc * (SPT if scheme == 'rows1' else 1))
    st.sync()
    n = (8 if size >= 4096 else 40) * cyc * (SPT if scheme == 'rows1' else 1)
    ms, launches = st.time_steps(n)
    per_tick = ms * 1e3 / n * (SPT if scheme == 'rows1' else 1)
    plan = st.launch_plan()
    st.close()
    return per_tick, m, g, H, plan


def main():
    print('predicted scaling of `bench.py --gpus N` (Fenton 4v, fast policy); kernels measured on one MI355X, exchange modelled')
    for size, label in ((4096, 'BASELINE configs[3]: 4096x4096, strong scaling'), (512, "north_star's 512x512 at 1/2/4/8, strong scaling")):
        print('\n== %s' % label)
        base = None
        for world in (1, 2, 4, 8):
            for scheme in (('ghost',) if world == 1 else ('ghost', 'rows1')):
                t, m, g, H, plan = block_time(size, world, scheme)
                cells = size * size
                if world == 1:
                    base = t
                    print('N=1: %8.1f us per tick of kernels -> %7.0f Mcell-steps/s (plan K=%d x %d)' % (t, cells * SPT / t, plan[0], plan[1]))
                    continue
                halo_arrays = 4 if scheme == 'ghost' else 1
                msg = g * halo_arrays * size * 4                      # bytes per neighbour and direction
                n_ex = 1.0 / m if scheme == 'ghost' else SPT          # exchanges per tick
                for path, lat in LAT_US.items():
                    ex = lat + msg / (XGMI_GBS * 1e3)                 # us per exchange
                    tall = (size // world) * ((size + 43) // 44) // 25 >= 4 * 256
                    tick = max(t, n_ex * ex + 0.15 * t) if tall else t + n_ex * ex
                    print('N=%d %-5s halo every %s, %3d ghost rows, slab %4d rows: kernels %8.1f us per tick; exchange %6.1f us x %.2f per tick (%s); '
                          'predicted tick %8.1f us -> %7.0f Mcell-steps/s, efficiency %.2f'
                          % (world, scheme, ('%d ticks' % m) if scheme == 'ghost' else 'sub-step', g, H, t, ex, n_ex, path, tick, cells * SPT / tick,
                             base / (world * tick)))


if __name__ == '__main__':
    main()
