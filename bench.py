#!/usr/bin/env python3
"""bench.py — million cell-steps/s of the fused stencil+reaction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model fenton|br|court] [--size S] ...

A "step" here is one run() TICK of the reference driver = `dt_per_step` explicit sub-steps of the
whole grid (Fenton 10, fenton.py:133-138; BR 5; Courtemanche 1), i.e. one `sess.run(_ode_op)`
(ionic.py:203).  The metric counts SUB-steps:  value = H*W * K*dt_per_step / wall_s / 1e6.

N = 1 (default): BASELINE.json configs[1] — Fenton 4v, 512x512, dt 0.1, diff 1.5, hole (256,256,30),
S1 column, S2 'luq' at tick 210 — driven tick by tick exactly as `IonicModel.run()` drives it.
N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is sharded by row blocks
(fib_tf_amd/sharded.py), halos travel as RCCL point-to-point; weak scaling: every rank owns a
512-row x W block (`--rows-per-gpu`), W = `--size`.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the oracle = CPU restatement of the reference, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'traffic_r01.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
# algorithmic bytes per cell per sub-step: every state array read once + every updated
# array written once, float32, + 4 B for the phase field
ALGO_BYTES = {'fenton': 32, 'br': 64, 'court': 106.8}


def make_model(args, height=None, device=0):
    from fib_tf_amd.fenton import Fenton4v
    from fib_tf_amd.br import BeelerReuter
    from fib_tf_amd.court import Courtemanche
    S = args.size
    H = height or S
    sc = S / 512.0
    base = {'width': S, 'height': H, 'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000, 'timeline': False,
            'timeline_name': 'timeline.json', 'save_graph': False, 'device': device, 'fast_math': not args.exact,
            'halo_ticks': getattr(args, 'halo_ticks', 4)}
    if args.model == 'fenton':                        # fenton.py:156-171
        m = Fenton4v(dict(base, diff=1.5))
        m.add_hole_to_phase_field(256 * sc, H / 2.0, 30 * sc)
        s2 = ('luq', 1.0, 210)
    elif args.model == 'br':                          # br.py:348-371 (cheby=True, skip=False)
        m = BeelerReuter(dict(base, diff=0.809, cheby=not args.no_cheby, skip=args.skip))
        m.add_hole_to_phase_field(150 * sc, 200 * sc * H / S, 40 * sc)
        s2 = ('luq', 10.0, 300)
    else:                                             # court.py:586-611
        m = Courtemanche(dict(base, diff=0.809))
        m.add_hole_to_phase_field(256 * sc, H / 2.0, 30 * sc)
        m.add_hole_to_phase_field(256 * sc, H / 2.0, 250 * sc, neg=True)
        s2 = ('luq', 10.0, 350)
    return m, s2


def cpu_baseline(args, seconds=12.0):
    """the oracle (CPU restatement, OpenMP over all host cores) on a bounded sample of the SAME
    workload: same grid, phase field and initial conditions, fewer sub-steps"""
    import oracle
    oracle.build()
    m, _ = make_model_host(args)
    slab, phi, run = m
    run(slab, 20)                                     # page in + calibrate
    t0 = time.perf_counter()
    run(slab, 40)
    per = (time.perf_counter() - t0) / 40
    n = int(max(40, min(seconds / per, 200000)))
    t0 = time.perf_counter()
    run(slab, n)
    dt = time.perf_counter() - t0
    cells = slab.shape[1] * slab.shape[2]
    threads = oracle.num_threads()
    # one-thread figure (a scalar port of the reference), about 3 s
    oracle.set_threads(1)
    n1 = int(max(10, min(3.0 / (per * threads * 0.6), 20000)))
    t0 = time.perf_counter()
    run(slab, n1)
    dt1 = time.perf_counter() - t0
    oracle.set_threads(threads)
    model_name = ''
    try:
        with open('/proc/cpuinfo') as f:
            model_name = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), '')
    except OSError:
        pass
    return {'value': round(cells * n / dt / 1e6, 2), 'unit': 'Mcell-steps/s', 'cores': threads,
            'kind': 'port', 'sample': '%d sub-steps of the same %dx%d %s workload (%.1f s), oracle/fib_oracle.c '
            'with OpenMP' % (n, slab.shape[1], slab.shape[2], args.model, dt),
            'value_1thread': round(cells * n1 / dt1 / 1e6, 2), 'cpu_model': model_name,
            'host_cpu_count': os.cpu_count(), 'omp_num_threads': os.environ.get('OMP_NUM_THREADS', 'unset')}


def make_model_host(args):
    """host-only construction of the benchmark workload for the oracle (no GPU calls)"""
    import oracle
    from fib_tf_amd.ionic import IonicModel
    S = args.size
    sc = S / 512.0
    g = IonicModel({'width': S, 'height': S})
    if args.model == 'fenton':
        g.add_hole_to_phase_field(256 * sc, S / 2.0, 30 * sc)
        slab = np.zeros((4, S, S), np.float32)
        slab[1] = 1.0
        slab[2] = 1.0
        slab[0][:, 1] = 1.0
        run = lambda s, n: oracle.fenton_run(s, 0.1, 1.5, g.phase, n)
    elif args.model == 'br':
        from fib_tf_amd.br import BeelerReuter
        g.add_hole_to_phase_field(150 * sc, 200 * sc, 40 * sc)
        slab = np.empty((8, S, S), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            slab[i] = v
        slab[0][:, 1] = 10.0
        b = BeelerReuter({'width': 8, 'height': 8, 'dt': 0.1, 'diff': 0.809})
        tbl = None if args.no_cheby else b.chebyshev_table().astype(np.float32)
        run = lambda s, n: oracle.br_run(s, 0.1, 0.809, g.phase, tbl, args.skip, max(1, n // 5))
    else:
        from fib_tf_amd.court import INITIAL
        g.add_hole_to_phase_field(256 * sc, S / 2.0, 30 * sc)
        g.add_hole_to_phase_field(256 * sc, S / 2.0, 250 * sc, neg=True)
        slab = np.empty((21, S, S), np.float32)
        for i, (_, v) in enumerate(INITIAL):
            slab[i] = v
        slab[0][:, :25] = 20.0
        run = lambda s, n: oracle.court_run(s, 0.1, 0.809, g.phase, True, 0, n)
    return (slab, g.phase, run), None


def bench_single(args):
    m, (loc, amp, s2_ms) = make_model(args, device=0)
    m.define()
    m.add_pace_op('s2', loc, amp)
    s2 = m.millisecond_to_step(s2_ms)
    st = m._stepper
    fused, per_tick = st.launch_plan()
    spt = m.dt_per_step
    court = args.model == 'court'
    tick = 0

    def advance(n):
        nonlocal tick
        for _ in range(n):                            # exactly IonicModel.run()'s loop body
            st.step(1)
            if court and tick % 10 == 0:              # court.py:615-617
                st.step_slow()
            if tick == s2:
                m.fire_op('s2')
            tick += 1

    st.step(1)                                        # setup, not a warm-up step: loads the code object
    st.sync()
    tick = 1
    advance(args.warmup)
    st.sync()
    t0 = time.perf_counter()
    advance(args.steps)
    st.sync()
    wall = time.perf_counter() - t0

    cells = m.height * m.width
    value = cells * args.steps * spt / wall / 1e6
    # dominant kernel, HIP events on the kernel's own stream, back-to-back launches
    ms, launches = st.time_steps(max(50, min(args.steps, 500)))
    us_per_launch = ms * 1000.0 / launches
    abytes = ALGO_BYTES[args.model] + (4 if m.phase is not None else 0)
    achieved = abytes * cells * fused / (us_per_launch * 1e-6) / 1e9
    traffic = None
    try:                                              # measured offline (PMC cannot run inside the bench)
        key = '%s/%s/%dx%d/K%d' % (args.model, 'exact' if args.exact else 'fast', m.height, m.width, fused)
        traffic = json.load(open(TRAFFIC_JSON)).get(key, {}).get('hbm_bytes_per_launch_corrected')
    except (OSError, ValueError):
        pass
    out = {
        'metric': 'million cell-steps/sec (grid_cells x timesteps / wall_s), %s %dx%d' % (
            {'fenton': '4v', 'br': 'BR', 'court': 'Courtemanche'}[args.model], m.height, m.width),
        'value': round(value, 1), 'unit': 'Mcell-steps/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(wall * 1000.0 / args.steps, 6), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s %dx%d, dt=0.1 ms, phase-field hole, S1 + S2 pacing (BASELINE configs[%d]); '
                               '1 step = 1 run() tick = %d sub-steps' % (
                                   args.model, m.height, m.width, {'fenton': 1, 'br': 2, 'court': 4}[args.model], spt),
                   'sub_steps_per_tick': spt, 'fused_sub_steps_per_launch': fused, 'launches_per_tick': per_tick,
                   'arithmetic': 'exact (one rounding per reference op)' if args.exact else 'fast_math (default policy)',
                   'parallelism': 'single device'},
        'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                     'kernel': '%s<%s, K=%d>' % ('strip_kernel' if fused > 1 and args.model in ('fenton', 'br') else 'tick_kernel', args.model, fused), 'us_per_launch': round(us_per_launch, 3),
                     'algorithmic_bytes_per_launch': int(abytes * cells * fused),
                     'note': 'working set is LDS/L2/Infinity-Cache resident; algorithmic bytes are what a '
                             'one-step-per-pass implementation must move, K fused sub-steps move them once'},
    }
    try:                                              # achievable-bandwidth yardstick, measured in this very run
        from fib_tf_amd import _lib
        out['roofline']['copy_bandwidth_measured'] = round(_lib.copy_bandwidth(1 << 30, 5, m.device), 1)
    except Exception as e:                            # never lose the result line over the yardstick
        out['roofline']['copy_bandwidth_measured'] = None
        print('copy bandwidth not measured: %s' % e, file=sys.stderr)
    if not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(args)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000, help='ticks timed (1 tick = dt_per_step sub-steps)')
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--model', default='fenton', choices=['fenton', 'br', 'court'])
    ap.add_argument('--size', type=int, default=0, help='grid width (and height at N=1); default 512 (1024 court)')
    ap.add_argument('--rows-per-gpu', type=int, default=512)
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='N>1: weak = rows-per-gpu rows on every GPU (default); strong = the fixed size x size grid '
                         'split over the N GPUs (north_star\'s "512x512 at 1/2/4/8")')
    ap.add_argument('--exact', action='store_true', help="config['fast_math']=False: one rounding per reference op")
    ap.add_argument('--no-cheby', action='store_true')
    ap.add_argument('--skip', action='store_true')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--halo-ticks', type=int, default=4, help='N > 1: ticks between two halo exchanges (ghost zone depth)')
    args = ap.parse_args()
    if not args.size:
        args.size = 1024 if args.model == 'court' else 512
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 or world > 1:
        from fib_tf_amd.sharded import bench_sharded
        out = bench_sharded(args, make_model, cpu_baseline, ALGO_BYTES, HBM_PEAK_GBS)
        if out is None:
            return
    else:
        out = bench_single(args)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
