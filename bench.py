#!/usr/bin/env python3
"""bench.py — million cell-steps/s of the fused stencil+reaction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model fenton|br|court] [--size S] ...

A "step" here is one run() TICK of the reference driver = `dt_per_step` explicit sub-steps of the
whole grid (Fenton 10, fenton.py:133-138; BR 5; Courtemanche 1), i.e. one `sess.run(_ode_op)`
(ionic.py:203).  The metric counts SUB-steps:  value = H*W * K*dt_per_step / wall_s / 1e6.

N = 1 (default): BASELINE.json configs[1] — Fenton 4v, 512x512, dt 0.1, diff 1.5, hole (256,256,30),
S1 column, S2 'luq' at tick 210 — driven tick by tick exactly as `IonicModel.run()` drives it.

N > 1: BASELINE.json configs[3] — Fenton 4v 4096x4096 split into N row blocks (fib_tf_amd/sharded.py),
hole (2048,2048,240), halos as RCCL point-to-point; the grid is fixed, so the N > 1 lines are a strong-
scaling series (`--scaling strong --size 512` = north_star's "512x512 at 1/2/4/8"; `--scaling weak` =
`--rows-per-gpu` rows on every GPU).  One process per GPU.  Under a launcher (WORLD_SIZE set: `python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) this process IS one rank.  Without one,
`python bench.py --gpus N` starts the N ranks itself as child processes (the parent makes no GPU call),
relays rank 0's JSON line, and exits non-zero if a rank fails or the box has fewer than N devices.

Timing protocol: untimed set-up ticks (code objects, RCCL channels, clocks), W warm-up ticks, then
`--repeats` (3) timed regions of EXACTLY K ticks each, every one bracketed by a barrier + device synchronisation;
per region the MAX over ranks, over regions the MEDIAN.  Prints ONE JSON line (rank 0) with `roofline`
(dominant kernel, HIP-event timed on its own stream; + `issue` = the instruction-issue ceiling and `cache` = L2
hit rate from the committed rocprofv3 counters), `cpu_baseline` (the oracle = CPU restatement of the reference,
timed on this host's cores), `value_with_snapshots` (the reference driver's read-backs inside the timed region)
and `exact` (the rounding-faithful arithmetic policy timed the same way).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# rocprofv3 --pmc passes of this command, condensed by tools/prof_summary.py (PMC cannot run inside the bench)
COUNTERS_JSON = [os.path.join(ROOT, 'profiles', 'counters_r04.json'), os.path.join(ROOT, 'profiles', 'counters_r03.json'), os.path.join(ROOT, 'profiles', 'counters_r02.json'),
                 os.path.join(ROOT, 'profiles', 'traffic_r01.json')]
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)
N_SIMD, CLOCK_GHZ = 1024, 2.4    # 256 CUs x 4 SIMD-32; max shader clock (MI355X_MICROARCH.md, chip-level parameters)
CYC_VALU, CYC_TRANS = 2.0, 8.0   # issue cycles of a wave64 VALU / transcendental instruction on one SIMD (same guide)
# algorithmic bytes per cell per sub-step: every state array read once + every updated
# array written once, float32, + 4 B for the phase field.  Courtemanche: the fast tick reads 21 arrays and
# writes 4 (100 B); the every-10th tick also assigns the other 17 (168 B); mean 106.8.
ALGO_BYTES = {'fenton': 32, 'br': 64, 'court': 106.8}
COURT_FAST_BYTES, COURT_SLOW_BYTES = 100.0, 168.0


def make_model(args, height=None, device=0, exact=None):
    from fib_tf_amd.fenton import Fenton4v
    from fib_tf_amd.br import BeelerReuter
    from fib_tf_amd.court import Courtemanche
    S = args.size
    H = height or S
    sc = S / 512.0
    exact = args.exact if exact is None else exact
    base = {'width': S, 'height': H, 'dt': 0.1, 'dt_per_plot': 10, 'duration': 1000, 'timeline': False,
            'timeline_name': 'timeline.json', 'save_graph': False, 'device': device, 'fast_math': not exact,
            'halo_ticks': getattr(args, 'halo_ticks', 0), 'halo': getattr(args, 'halo', None)}
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


def cpu_baseline(args, height=None, seconds=12.0):
    """the oracle (CPU restatement, OpenMP) on a bounded sample of the SAME workload: same grid, phase field and
    initial conditions, fewer sub-steps.  Three figures: the default thread count (the GPU box's stated CPU share,
    16), every core this process may use (affinity + cgroup quota), and one thread."""
    import oracle
    oracle.build()
    slab, phi, run = make_model_host(args, height)
    cells = slab.shape[1] * slab.shape[2]
    threads = oracle.num_threads()
    run(slab, 10)                                     # page in + calibrate
    t0 = time.perf_counter()
    run(slab, 20)
    per = (time.perf_counter() - t0) / 20
    n = int(max(20, min(seconds / per, 200000)))
    t0 = time.perf_counter()
    run(slab, n)
    dt = time.perf_counter() - t0
    out = {'value': round(cells * n / dt / 1e6, 2), 'unit': 'Mcell-steps/s', 'cores': threads, 'kind': 'port',
           'sample': '%d sub-steps of the same %dx%d %s workload (%.1f s), oracle/fib_oracle.c with OpenMP'
                     % (n, slab.shape[1], slab.shape[2], args.model, dt)}
    # every usable core ("all host cores"), about 5 s
    allc = oracle.usable_cores()
    if allc != threads:
        oracle.set_threads(allc)
        run(slab, 10)
        na = int(max(20, min(5.0 / (per * threads / allc), 400000)))
        t0 = time.perf_counter()
        run(slab, na)
        dta = time.perf_counter() - t0
        out['value_all_cores'] = round(cells * na / dta / 1e6, 2)
    else:
        out['value_all_cores'] = out['value']
    out['cores_all'] = allc
    # one-thread figure (a scalar port of the reference), about 3 s
    oracle.set_threads(1)
    n1 = int(max(5, min(3.0 / (per * threads * 0.6), 20000)))
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
    out.update({'value_1thread': round(cells * n1 / dt1 / 1e6, 2), 'cpu_model': model_name,
                'host_cpu_count': os.cpu_count(), 'omp_num_threads': os.environ.get('OMP_NUM_THREADS', 'unset')})
    return out


def make_model_host(args, height=None):
    """host-only construction of the benchmark workload for the oracle (no GPU calls)"""
    import oracle
    from fib_tf_amd.ionic import IonicModel
    S = args.size
    H = height or S
    sc = S / 512.0
    g = IonicModel({'width': S, 'height': H})
    if args.model == 'fenton':
        g.add_hole_to_phase_field(256 * sc, H / 2.0, 30 * sc)
        slab = np.zeros((4, H, S), np.float32)
        slab[1] = 1.0
        slab[2] = 1.0
        slab[0][:, 1] = 1.0
        run = lambda s, n: oracle.fenton_run(s, 0.1, 1.5, g.phase, n)
    elif args.model == 'br':
        from fib_tf_amd.br import BeelerReuter
        g.add_hole_to_phase_field(150 * sc, 200 * sc * H / S, 40 * sc)
        slab = np.empty((8, H, S), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            slab[i] = v
        slab[0][:, 1] = 10.0
        b = BeelerReuter({'width': 8, 'height': 8, 'dt': 0.1, 'diff': 0.809})
        tbl = None if args.no_cheby else b.chebyshev_table().astype(np.float32)
        run = lambda s, n: oracle.br_run(s, 0.1, 0.809, g.phase, tbl, args.skip, max(1, n // 5))
    else:
        from fib_tf_amd.court import INITIAL
        g.add_hole_to_phase_field(256 * sc, H / 2.0, 30 * sc)
        g.add_hole_to_phase_field(256 * sc, H / 2.0, 250 * sc, neg=True)
        slab = np.empty((21, H, S), np.float32)
        for i, (_, v) in enumerate(INITIAL):
            slab[i] = v
        slab[0][:, :25] = 20.0
        run = lambda s, n: oracle.court_run(s, 0.1, 0.809, g.phase, True, 0, n)
    return slab, g.phase, run


def counters_for(key):
    """the committed rocprofv3 --pmc record of this kernel shape, or {}"""
    for path in COUNTERS_JSON:
        try:
            rec = json.load(open(path)).get(key)
        except (OSError, ValueError):
            rec = None
        if rec:
            return dict(rec, source=os.path.relpath(path, ROOT))
    return {}


def roofline_extras(roof, key, us_per_launch, own_time=False, tile=None, ticks_per_launch=1):
    """traffic / issue ceiling / cache hit rates of the dominant kernel from the committed counter passes;
    own_time: the timed launches are a mix of kernels (Courtemanche), so the counters of the dominant one are set against
    its own duration in the same profile instead of the mean launch of the mix.
    A record made of multi-tick launches holds PER-TICK figures (`per: tick`); they are scaled to this run's ticks per
    launch.  A record whose kernel has another tile shape than this run's plan (first-tick measurement picks per box) is
    not mixed in: it is named, flagged `stale`, and no figure is derived from it."""
    rec = counters_for(key)
    roof['counters_key'] = key
    if rec and tile is not None and rec.get('tile') and list(rec['tile']) != list(tile):
        roof['counters_stale'] = ('%s [%s] was recorded on tile %s, this run chose %s: no traffic / issue / cache figures'
                                  % (rec.get('source'), key, 'x'.join(map(str, rec['tile'])), 'x'.join(map(str, tile))))
        return roof
    scale = float(ticks_per_launch) if rec.get('per') == 'tick' else 1.0
    if own_time and rec.get('us_per_launch_under_trace'):
        us_per_launch = rec['us_per_launch_under_trace']
    t = rec.get('hbm_bytes_per_launch_corrected')
    roof['traffic'] = None if t is None else t * scale
    if t is not None and us_per_launch > 0:           # what the launch really pulls from / pushes to HBM, against the peak
        roof['hbm_traffic_frac'] = round(t * scale / (us_per_launch * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    valu, trans = rec.get('SQ_INSTS_VALU'), rec.get('SQ_INSTS_VALU_TRANS_F32')
    if valu is not None:
        # wave-instructions per launch over all waves; a SIMD issues a wave64 VALU instruction in 2 cycles and a
        # transcendental in 8: the time the chip's 1024 SIMDs need just to ISSUE them, against the launch time
        valu *= scale
        t = (trans or 0.0) * scale
        cycles = ((valu - t) * CYC_VALU + t * CYC_TRANS) / N_SIMD
        us = cycles / (CLOCK_GHZ * 1e3)
        roof['issue'] = {'bound': 'valu-issue', 'valu_wave_instr_per_launch': round(valu), 'trans_wave_instr_per_launch':
                         None if trans is None else round(t), 'salu_wave_instr_per_launch':
                         None if rec.get('SQ_INSTS_SALU') is None else round(rec['SQ_INSTS_SALU'] * scale),
                         'cycles_per_valu': CYC_VALU, 'cycles_per_trans': CYC_TRANS, 'simds': N_SIMD, 'clock_ghz': CLOCK_GHZ,
                         'issue_us_per_launch': round(us, 3), 'frac': round(us / us_per_launch, 4),
                         'note': 'fraction of the launch the SIMDs spend issuing vector instructions; this, not HBM, is '
                                 'the ceiling once K sub-steps are fused (frac of the HBM figure can then exceed 1)'}
    hit, miss = rec.get('TCC_HIT_sum'), rec.get('TCC_MISS_sum')
    if hit is not None and miss is not None and hit + miss > 0:
        roof['cache'] = {'l2_hit_rate': round(hit / (hit + miss), 4), 'TCC_HIT_sum': round(hit * scale), 'TCC_MISS_sum': round(miss * scale)}
        for k in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_DRAM_sum', 'TCC_EA0_WRREQ_sum', 'TCC_EA0_WRREQ_DRAM_sum'):
            if rec.get(k) is not None:
                roof['cache'][k] = round(rec[k] * scale)
        roof['cache']['note'] = ('TCC_HIT/MISS: the XCDs\' L2s.  TCC_EA0_*REQ: 64-B requests leaving the L2s towards the '
                                 'fabric; on gfx950 the _DRAM variants count the same requests (Infinity-Cache hits are not '
                                 'told apart, MI355X_MICROARCH.md "HBM"), so no Infinity-Cache hit rate can be derived; the '
                                 'working set of this launch is far below its 256 MiB')
    if rec:
        roof['counters_source'] = '%s [%s], kernel %s%s' % (rec.get('source'), key, rec.get('kernel', '?'),
                                                            ' (recorded per tick, scaled to %d ticks per launch)' % ticks_per_launch
                                                            if scale != 1.0 else '')
    return roof


def driver_loop(m, st, s2, court):
    """IonicModel.run()'s loop body with the reference driver's side calls (court.py:612-617, fenton.py:181-183);
    `snap` adds its read-backs (fenton.py:184-185: image() every 10 ms; court.py:618: the trend probe)"""
    state = {'tick': 0}
    ds = max(1, m.millisecond_to_step(10))

    def advance(n, snap=False):
        tick = state['tick']
        if not snap:
            st.expect(n)                              # as run() does without a screen: the whole loop is one series
        else:                                         # with frames: the ticks up to the first one (run(im) knows its period)
            st.expect(min(n, (-tick) % ds + 1))
        for k in range(n):
            st.step(1)
            if court and tick % 10 == 0:
                st.step_slow()
                if snap:
                    m.fire_op('trend')
            if tick == s2:
                m.fire_op('s2')
            if snap and tick % ds == 0:
                st.expect(min(ds, n - 1 - k))         # as run(im) does before each frame: the ticks up to the next one
                m.image()
            tick += 1
        state['tick'] = tick
    return advance


def timed_regions(advance, sync, steps, repeats, barrier=None, snap=False, launches=None):
    """`repeats` regions of exactly `steps` ticks, each bracketed by barrier + device sync; wall seconds per region
    (`launches`: a callable counting kernel launches so far; its increments per region are appended to `launches.log`)"""
    walls = []
    for _ in range(repeats):
        sync()
        if barrier:
            barrier()
        l0 = launches() if launches else 0
        t0 = time.perf_counter()
        advance(steps, snap)
        sync()
        if barrier:
            barrier()
        walls.append(time.perf_counter() - t0)
        if launches:
            launches.log.append(launches() - l0)
    return walls


def kernel_key(args, exact, H, W, fused, shard=False, ticks=1):
    return '%s/%s/%dx%d/%s%d%s' % (args.model, 'exact' if exact else 'fast', H, W, 'T' if ticks > 1 else 'K',
                                   ticks if ticks > 1 else fused, '/shard' if shard else '')


def measure_single(args, exact, with_extras, snapshots=None, height=None, device=0):
    """one model on one device: (value, ms_per_tick, roofline dict, walls, snapshots value, model)"""
    snapshots = with_extras if snapshots is None else snapshots
    m, (loc, amp, s2_ms) = make_model(args, height=height, device=device, exact=exact)
    m.define()
    m.add_pace_op('s2', loc, amp)
    s2 = m.millisecond_to_step(s2_ms)
    st = m._stepper
    fused, per_tick = st.launch_plan()
    spt = m.dt_per_step
    court = args.model == 'court'
    advance = driver_loop(m, st, s2, court)
    st.step(1)                                        # set-up, not a warm-up step: loads the code object
    st.sync()
    # the interpreter's cyclic garbage collector: a full collection of this process takes ~40 ms (measured: one image() in
    # ~1350 took 40 ms instead of 0.16, always the same one, none with the collector off) — a third of a 5000-tick region.
    # Everything allocated so far is moved out of the collector's sight; what the timed loops allocate is freed by
    # reference counting.
    import gc
    gc.collect()
    gc.freeze()
    advance(args.setup)                               # set-up: clocks and caches in their steady state
    advance(args.warmup)
    def nlaunch():
        return st.launch_stats()['launches']
    nlaunch.log = []
    walls = timed_regions(advance, st.sync, args.steps, args.repeats, launches=nlaunch)
    wall = statistics.median(walls)
    m._launches_per_region = nlaunch.log
    cells = m.height * m.width
    value = cells * args.steps * spt / wall / 1e6

    # dominant kernel: HIP events on the kernel's own stream, back-to-back launches.  Courtemanche: around the SAME tick
    # mix as above (every 10th tick's fused fast+slow launch included).  Multi-tick launches (Fenton / Beeler-Reuter on a
    # grid whose tiles are all resident at once): whole launches of `tpl` ticks, handed over in one call.
    tpl = st.ticks_per_launch()
    multi = tpl > 1 and not court
    nt = max(50, min(args.steps, 500))
    if court:
        nt = (nt + 9) // 10 * 10                      # whole fast/slow periods
    st.sync()
    if multi:
        nt = max(4, min(16, args.steps // tpl)) * tpl
        st.time_begin()
        st.step(nt)
        ms, launches = st.time_end()
    else:
        st.time_begin()
        advance(nt)
        ms, launches = st.time_end()
    us_per_launch = ms * 1000.0 / max(1, launches)
    abytes = ALGO_BYTES[args.model] + (4 if m.phase is not None else 0)
    per_launch_bytes = abytes * cells * spt * nt / max(1, launches)      # mean over the launches of the mix
    achieved = per_launch_bytes / (us_per_launch * 1e-6) / 1e9
    name = 'strip_kernel' if fused > 1 and args.model in ('fenton', 'br') else 'tick_kernel'
    if multi:
        kname = 'strip_mt_kernel<%s, K=%d, %d ticks per launch>' % (args.model, fused, nt // max(1, launches))
    elif tpl > 1:
        kname = 'strip_kernel<%s on aggregates, %d ticks per launch>' % (args.model, tpl)
    else:
        kname = '%s<%s, K=%d>' % (name, args.model, fused)
    roof = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
            'kernel': kname, 'us_per_launch': round(us_per_launch, 3),
            'launches_timed': launches, 'ticks_timed': nt, 'ticks_per_launch': round(nt / max(1, launches), 2),
            'us_per_tick': round(ms * 1000.0 / nt, 3),
            'algorithmic_bytes_per_launch': int(per_launch_bytes),
            'note': 'working set is LDS/L2/Infinity-Cache resident; algorithmic bytes are what a '
                    'one-step-per-pass implementation must move, K fused sub-steps move them once'}
    if multi:
        roof['note'] += ('; one launch advances %d ticks: a tile keeps its cells in registers and re-reads the rim of its '
                         'compute box from its eight neighbours between two ticks (the state arrays are read at the first '
                         'and written at the last tick of a launch)' % (nt // max(1, launches)))
        roof['frac_is'] = ('the ALGORITHMIC bytes of SURVEY 8d (what one pass per sub-step must move) per launch time over the HBM '
                           'peak — an equivalent rate, not this launch\'s HBM traffic (`traffic`, `hbm_traffic_frac`), which is ~10x '
                           'lower: the launch is bound by instruction issue (`issue`) and its tick boundaries, not by HBM')
    if court:
        roof['note'] += ('; Courtemanche: the timed mix is 9 fast ticks (%.0f B/cell) + 1 fused fast+slow tick '
                         '(%.0f B/cell) per 10, bytes and time both of the mix' % (COURT_FAST_BYTES + 4, COURT_SLOW_BYTES + 4))
        if tpl > 1:
            roof['note'] += ('; fast policy: the 9 fast ticks run as 3 launches of %d ticks, temporally blocked, on five '
                             'per-cell aggregates of the slow variables (12 arrays read instead of 16): the launches move '
                             'fewer bytes than the algorithmic figure, which is why frac can exceed 1 — see `issue`' % tpl)
    snaps = None
    if with_extras:
        tw, th, tr = st.plan_tile()
        roofline_extras(roof, kernel_key(args, exact, m.height, m.width, fused, ticks=tpl), us_per_launch, own_time=court,
                        tile=(tw, th, tr), ticks_per_launch=max(1, nt // max(1, launches)) if multi else 1)
    if snapshots:
        m.image()                                     # set-up: the pinned staging buffer of the read-backs
        def nlaunch2():
            return st.launch_stats()['launches']
        nlaunch2.log = []
        ws = timed_regions(advance, st.sync, args.steps, args.repeats, snap=True, launches=nlaunch2)   # regions and median like `value`
        snaps = cells * args.steps * spt / statistics.median(ws) / 1e6
        m._snap_regions = {'wall_ms_per_region': [round(w * 1e3, 4) for w in ws], 'launches_per_region': nlaunch2.log}
    return value, wall * 1000.0 / args.steps, roof, walls, snaps, m


def plan_text(st):
    fused, per_tick = st.launch_plan()
    t = st.plan_tile()
    tpl = st.ticks_per_launch()
    return '%d sub-steps per launch x %d launch(es) per tick%s, tile %dx%d (%s)' % (
        fused, per_tick, (', up to %d ticks per launch' % tpl) if tpl > 1 else '', t[0], t[1],
        ('%d rows per wave' % t[2]) if t[2] > 0 else ('%d threads' % -t[2]))


# the other single-GPU configurations of BASELINE.json, timed in the same run with a bounded number of ticks
CONFIG_LEGS = [
    ('configs[2] Beeler-Reuter 8-var 512x512, cheby=True', 'br', 512, 600, 200),
    ('configs[4] Courtemanche 21-var 1024x1024, slow fired every 10th tick as court.py:612-617 does', 'court', 1024, 600, 200),
    ('configs[3] grid on ONE device: Fenton 4v 4096x4096', 'fenton', 4096, 100, 30),
]


def config_legs(args):
    import copy
    out = []
    for label, model, size, steps, setup in CONFIG_LEGS:
        a = copy.copy(args)
        a.model, a.size, a.steps, a.setup, a.warmup, a.repeats, a.exact = model, size, steps, setup, 20, 3, False
        try:
            value, ms_tick, roof, walls, _, m = measure_single(a, False, True, snapshots=False)
            st = m._stepper
            leg = {'config': label, 'value': round(value, 1), 'unit': 'Mcell-steps/s', 'ms_per_step': round(ms_tick, 6),
                   'steps': steps, 'timing': 'median of 3 regions of %d ticks' % steps, 'plan': plan_text(st),
                   'roofline': {k: roof.get(k) for k in ('frac', 'achieved', 'us_per_launch', 'us_per_tick', 'ticks_per_launch', 'kernel',
                                                         'traffic', 'counters_key', 'counters_source', 'counters_stale')
                                if roof.get(k) is not None},
                   'issue_frac': roof.get('issue', {}).get('frac')}
            st.close()
        except Exception as e:                            # a leg never costs the headline line
            leg = {'config': label, 'error': '%s: %s' % (type(e).__name__, e)}
        out.append(leg)
    return out


def bench_single(args):
    value, ms_tick, roof, walls, snaps, m = measure_single(args, args.exact, True)
    st = m._stepper
    fused, per_tick = st.launch_plan()
    spt = m.dt_per_step
    out = {
        'metric': 'million cell-steps/sec (grid_cells x timesteps / wall_s), %s %dx%d' % (
            {'fenton': '4v', 'br': 'BR', 'court': 'Courtemanche'}[args.model], m.height, m.width),
        'value': round(value, 1), 'unit': 'Mcell-steps/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_tick, 6), 'higher_is_better': True, 'scaling': 'none (one device, nothing is scaled)',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s %dx%d, dt=0.1 ms, phase-field hole, S1 + S2 pacing (BASELINE configs[%d]); '
                               '1 step = 1 run() tick = %d sub-steps' % (
                                   args.model, m.height, m.width, {'fenton': 1, 'br': 2, 'court': 4}[args.model], spt),
                   'sub_steps_per_tick': spt, 'fused_sub_steps_per_launch': fused, 'launches_per_tick': per_tick,
                   'ticks_per_launch': st.ticks_per_launch(), 'launch_stats': st.launch_stats(),
                   'tile': '%dx%d cells, %s (chosen by measurement on the first tick)' % (
                       (lambda t: (t[0], t[1], ('%d rows per wave' % t[2]) if t[2] > 0 else ('%d threads' % -t[2])))(st.plan_tile())),
                   'arithmetic': 'exact (one rounding per reference op)' if args.exact else 'fast_math (default policy)',
                   'parallelism': 'single device', 'setup_ticks': args.setup + 1,
                   'timing': 'median of %d regions of %d ticks' % (args.repeats, args.steps)},
        'repeats': args.repeats, 'wall_ms_per_region': [round(w * 1e3, 4) for w in walls],
        'launches_per_region': getattr(m, '_launches_per_region', None),
        'roofline': roof,
        'value_with_snapshots': None if snaps is None else round(snaps, 1),
        'snapshots_regions': getattr(m, '_snap_regions', None),
        'snapshots_note': 'median of %d regions of the same %d ticks with the reference driver\'s read-backs inside the timed '
                          'region: image() (the potential, into page-locked memory) every 10 ms of simulated time '
                          '(fenton.py:184-185)' % (args.repeats, args.steps),
    }
    try:                                              # achievable-bandwidth yardstick, measured in this very run
        from fib_tf_amd import _lib
        out['roofline']['copy_bandwidth_measured'] = round(_lib.copy_bandwidth(1 << 30, 5, m.device), 1)   # (from the stock library)
    except Exception as e:                            # never lose the result line over the yardstick
        out['roofline']['copy_bandwidth_measured'] = None
        print('copy bandwidth not measured: %s' % e, file=sys.stderr)
    if not args.exact and not args.no_exact_leg:      # the rounding-faithful policy, timed the same way
        st.close()
        ev, ems, eroof, ewalls, _, em = measure_single(args, True, True, snapshots=False)
        ek, _ = em._stepper.launch_plan()
        out['exact'] = {'value': round(ev, 1), 'ms_per_step': round(ems, 6), 'arithmetic': 'exact (one float32 rounding '
                        'per reference op, no FMA contraction: the policy the bit-level parity claims hold for)',
                        'roofline_frac': eroof['frac'], 'us_per_launch': eroof['us_per_launch'],
                        'fused_sub_steps_per_launch': ek, 'plan': plan_text(em._stepper),
                        'traffic': eroof.get('traffic'), 'issue_frac': eroof.get('issue', {}).get('frac'),
                        'counters_source': eroof.get('counters_source', eroof.get('counters_stale'))}
        em._stepper.close()
    if args.model == 'fenton' and args.size == 512 and not args.exact and not args.no_config_legs:
        st.close()
        out['configs'] = config_legs(args)
    if not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(args)
    return out


# ---------------------------------------------------------------------------------------------------------
# N > 1
# ---------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_devices():
    """GPUs this process could open, WITHOUT initialising any runtime (the parent of the ranks stays off the GPU): the
    KFD topology's nodes with SIMDs, narrowed by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.  Falls back to asking a child process."""
    n = 0
    try:
        root = '/sys/class/kfd/kfd/topology/nodes'
        for d in os.listdir(root):
            with open(os.path.join(root, d, 'properties')) as f:
                props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except (OSError, ValueError):
        n = -1
    if n < 0:                                         # no sysfs view: a child counts and exits
        try:
            out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'], capture_output=True,
                                 text=True, timeout=300).stdout
            n = int(out.strip().splitlines()[-1])
        except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
            n = 0
    # a lease that exposes only some of the node's GPUs (device cgroup) still shows every node in sysfs: what this process can
    # OPEN is the render nodes it may read and write
    try:
        nodes = [f for f in os.listdir('/dev/dri') if f.startswith('renderD')]
        usable = sum(1 for f in nodes if os.access(os.path.join('/dev/dri', f), os.R_OK | os.W_OK))
        if nodes and n > usable:
            n = usable
    except OSError:
        pass
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes, one per GPU, relay
    rank 0's result line.  This process makes no GPU call and never re-executes itself."""
    one_device = os.environ.get('FIBTF_ONE_DEVICE') == '1'      # rehearsal: several gloo ranks on ONE GPU
    have = visible_devices()
    need = 1 if one_device else args.gpus
    if have < need:
        print('bench.py: --gpus %d needs %d HIP device(s), this box has %d' % (args.gpus, need, have), file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    # rank 0 writes its headline here before the side legs start: if a leg then takes a rank down (or all of them past the
    # time limit), the headline measured before is relayed instead of being lost
    import tempfile
    fd, headline_file = tempfile.mkstemp(prefix='fibtf_headline_', suffix='.json')
    os.close(fd)
    os.unlink(headline_file)
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', FIBTF_HEADLINE_FILE=headline_file)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()                                    # drain rank 0's stdout while it runs (a full pipe would block it)
    deadline = time.time() + args.spawn_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or all(c is not None for c in codes) or time.time() > deadline:
            if bad:
                rc = bad[0] if bad[0] > 0 else 1
                print('bench.py: a rank exited with code %d; stopping the others' % bad[0], file=sys.stderr)
            elif time.time() > deadline and any(c is None for c in codes):
                rc = 3
                print('bench.py: ranks still running after %d s; stopping them' % args.spawn_timeout, file=sys.stderr)
            break
        time.sleep(0.2)
    for p in procs:                                   # exactly the processes started above, nothing by pattern
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=15)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=15)
    out = ''.join(chunks)
    lines = [l for l in out.splitlines() if l.startswith('{')]
    if not lines and os.path.exists(headline_file):       # the ranks went down in a side leg: the headline stands
        try:
            early = json.load(open(headline_file))
            early['side_legs_note'] = ('a rank failed or ran out of time in a side leg (exit status %d); this is the headline '
                                       'rank 0 had written before the side legs started' % rc)
            out += json.dumps(early) + '\n'
            lines = [json.dumps(early)]
            rc = 0
        except (OSError, ValueError):
            pass
    if os.path.exists(headline_file):
        os.unlink(headline_file)
    if rc == 0 and not lines:
        print('bench.py: rank 0 printed no result line', file=sys.stderr)
        rc = 1
    for l in out.splitlines():                        # the result line to stdout, anything else a library printed to stderr
        (sys.stdout if l.startswith('{') else sys.stderr).write(l + '\n')
    sys.stdout.flush()
    return rc


def predicted_figure(size, world, scheme, transport):
    """the figure tools/predict_scaling.py wrote for this (grid, ranks, halo scheme, transport) — kernels of one rank's
    block measured on one MI355X, the exchange between two devices modelled — or None"""
    try:
        txt = open(os.path.join(ROOT, 'profiles', 'r03_predicted_scaling.txt')).read()
    except OSError:
        return None
    import re
    sect = None
    for line in txt.splitlines():
        if line.startswith('=='):
            sect = 4096 if '4096' in line else (512 if '512x512' in line else None)
        if sect != size:
            continue
        if world == 1:
            mm = re.match(r'N=1:.*->\s*(\d+) Mcell-steps/s', line)
            if mm:
                return {'value': float(mm.group(1)), 'source': 'profiles/r03_predicted_scaling.txt'}
            continue
        mm = re.match(r'N=(\d+) (\w+) halo.*kernels\s+([\d.]+) us per tick.*\((\w+)\); predicted tick\s+([\d.]+) us ->\s*(\d+) Mcell-steps/s', line)
        if mm and int(mm.group(1)) == world and mm.group(2) == scheme and mm.group(4) == transport:
            return {'value': float(mm.group(6)), 'kernels_us_per_tick': float(mm.group(3)), 'tick_us': float(mm.group(5)),
                    'transport_assumed': transport, 'source': 'profiles/r03_predicted_scaling.txt (kernels of one rank\'s block '
                    'measured on one MI355X, the exchange between two devices modelled)'}
    return None


def block_kernels(m, st, H, local, spt, halo_ticks):
    """rank 0's row block — owned + ghost rows, row offset, interleaved slab — as a stand-alone handle, its ticks timed with
    HIP events on its stream: the block's kernels without any exchange.  (us per launch, launches, ticks, fused sub-steps)"""
    from fib_tf_amd import _lib
    espt = getattr(st, 'eng_spt', spt)                  # sub-steps per engine tick: 1 under --halo rows1
    probe = _lib.Stepper(m.MODEL_ID, st.lh, m.width, m.dt, m.diff, flags=m._flags() | _lib.ROW_INTERLEAVED, device=local,
                         steps_per_tick=espt, global_height=H, row_offset=st.lo, ghost_top=st.gt, ghost_bottom=st.gb,
                         library=m._library)
    try:
        m._configure_stepper(probe)
        if m.phase is not None:
            probe.set_phase(np.ascontiguousarray(m.phase[st.lo:st.lo + st.lh]))
        probe.set_state(-1, st.eng.get_state(-1))
        probe.step(2 * halo_ticks)
        probe.sync()
        n = max(halo_ticks, 200 // halo_ticks * halo_ticks)
        ms, launches = probe.time_steps(n)
        return ms * 1000.0 / max(1, launches), launches, n * espt / float(spt), probe.launch_plan()[0]
    finally:
        probe.close()


def sharded_run(a, H, local, steps, repeats, setup, warmup):
    """one grid in row blocks over the ranks of the default process group, driven and timed as the contract says: set-up
    ticks, warm-up, `repeats` regions of exactly `steps` ticks between barrier + synchronisation, per region the max over
    ranks.  Every rank calls it; returns what the result line is made of (the model and its stepper stay open)."""
    import torch
    import torch.distributed as dist
    m, (loc, amp, s2_ms) = make_model(a, height=H, device=local)
    m.define()
    m.add_pace_op('s2', loc, amp)
    s2 = m.millisecond_to_step(s2_ms)
    st = m._stepper
    advance = driver_loop(m, st, s2, a.model == 'court')
    # (a one-rank group — only ever used to rehearse this path on one GPU — has a plain Stepper: no ghost zone)
    halo_ticks = getattr(st, 'halo_ticks', 1)
    advance(2 * halo_ticks)                             # set-up, not warm-up: two full exchange cycles create
    st.sync()                                           # the RCCL channels and load the code objects
    advance(setup // halo_ticks * halo_ticks)           # set-up: clocks in their steady state
    advance(warmup)

    def sync():
        st.sync()
        torch.cuda.synchronize()

    comm0 = getattr(st, 'comm_s', 0.0)
    walls = timed_regions(advance, sync, steps, repeats, barrier=dist.barrier)
    wt = torch.tensor(walls, dtype=torch.float64, device='cuda')
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)                       # per region: the slowest rank
    walls = [float(x) for x in wt.tolist()]
    comm = torch.tensor([getattr(st, 'comm_s', 0.0) - comm0], dtype=torch.float64, device='cuda')
    dist.all_reduce(comm, op=dist.ReduceOp.MAX)
    wall = statistics.median(walls)
    return {'m': m, 'st': st, 'advance': advance, 'sync': sync, 'walls': walls, 'wall': wall, 'halo_wait': float(comm.item()),
            'value': H * m.width * steps * m.dt_per_step / wall / 1e6, 'halo_ticks': halo_ticks,
            'setup_ticks': 2 * halo_ticks + setup // halo_ticks * halo_ticks}


def side_leg(a, H, local, rank, world, steps, repeats, setup, label):
    """a bounded run of another configuration on the same ranks (every rank calls it): value, whole tick, rank 0's kernels"""
    r = sharded_run(a, H, local, steps, repeats, setup, min(a.warmup, 8))
    m, st = r['m'], r['st']
    spt = m.dt_per_step
    kern = None
    if rank == 0 and hasattr(st, 'lh'):
        kern = block_kernels(m, st, H, local, spt, r['halo_ticks'])
    leg = None
    if rank == 0:
        scheme = getattr(st, 'halo_mode', 'ghost')
        transport = 'library' if getattr(st, 'rccl_direct', False) else 'torch'
        leg = {'config': label, 'value': round(r['value'], 1), 'unit': 'Mcell-steps/s', 'n_gpus': world,
               'ms_per_step': round(r['wall'] * 1000.0 / steps, 6), 'steps': steps,
               'timing': 'median of %d regions of %d ticks; per region the max over ranks' % (repeats, steps),
               'wall_ms_per_region': [round(w * 1e3, 4) for w in r['walls']],
               'whole_tick_us': round(r['wall'] * 1e6 / steps, 3), 'halo_scheme': scheme,
               'halo_transport': getattr(st, 'halo_path', 'none'), 'halo_ticks': r['halo_ticks'],
               'halo_wait_s_max_rank': round(r['halo_wait'], 4),
               'predicted': predicted_figure(a.size, world, scheme, transport) if a.model == 'fenton' and not a.exact else None}
        if kern:
            us_l, launches, nt, kf = kern
            leg['rank0_kernels_us_per_tick'] = round(us_l * launches / max(nt, 1e-9), 3)
            leg['rank0_block'] = '%d owned + %d ghost rows x %d, K=%d' % (st.rows, st.lh - st.rows, m.width, kf)
    st.sync()
    st.close()
    return leg


def bench_ranks(args):
    """this process is ONE rank of `world` (started by torch.distributed.run or by spawn_ranks)"""
    import copy
    import threading
    import torch
    import torch.distributed as dist
    from fib_tf_amd.sharded import init_from_env
    rank, world, local = init_from_env()
    strong = args.scaling == 'strong'
    H = args.size if strong else args.rows_per_gpu * world
    r = sharded_run(args, H, local, args.steps, args.repeats, args.setup, args.warmup)
    m, st, walls, wall = r['m'], r['st'], r['walls'], r['wall']
    spt = m.dt_per_step
    halo_ticks, ghost, halo_n = r['halo_ticks'], getattr(st, 'g', 0), getattr(st, 'halo_n', 0)
    nranks = torch.tensor([1], dtype=torch.int32, device='cuda')
    dist.all_reduce(nranks, op=dist.ReduceOp.SUM)                   # what the communicator itself counts
    fused, per_tick = st.launch_plan()
    backend = dist.get_backend()

    # per-GPU kernel figure (rank 0): the same row block as a stand-alone handle, HIP-event timed
    kern = None
    if rank == 0 and hasattr(st, 'lh'):
        kern = block_kernels(m, st, H, local, spt, halo_ticks)

    out = None
    if rank == 0:
        cells = H * m.width
        value = cells * args.steps * spt / wall / 1e6
        abytes = ALGO_BYTES[args.model] + (4 if m.phase is not None else 0)
        us_tick = wall * 1e6 / args.steps
        own = st.rows if hasattr(st, 'rows') else H
        tick_achieved = abytes * own * m.width * spt / (us_tick * 1e-6) / 1e9
        cfgi = 3 if (args.model == 'fenton' and strong and args.size == 4096) else None
        roof = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'traffic': None,
                'whole_tick': {'achieved': round(tick_achieved, 1), 'frac': round(tick_achieved / HBM_PEAK_GBS, 4),
                               'us_per_tick': round(us_tick, 3),
                               'note': 'rank 0\'s owned rows x algorithmic bytes / (wall / ticks): kernels + halo exchange'}}
        if kern:
            us_l, launches, nt, kf = kern
            per_launch = abytes * own * m.width * spt * nt / max(1, launches)
            ach = per_launch / (us_l * 1e-6) / 1e9
            roof.update({'achieved': round(ach, 1), 'frac': round(ach / HBM_PEAK_GBS, 4), 'us_per_launch': round(us_l, 3),
                         'kernel': '%s<%s, K=%d> on one rank\'s block (%d owned + %d ghost rows x %d)' % (
                             'strip_kernel' if kf > 1 else 'tick_kernel', args.model, kf, own, st.lh - own, m.width),
                         'algorithmic_bytes_per_launch': int(per_launch), 'launches_timed': launches,
                         'note': 'per GPU: the block\'s kernels alone (HIP events, no exchange; ghost-row work is not '
                                 'counted as useful bytes); whole_tick adds the halo exchange'})
            roofline_extras(roof, kernel_key(args, args.exact, st.lh, m.width, kf, shard=True), us_l)
        else:
            roof.update({'achieved': roof['whole_tick']['achieved'], 'frac': roof['whole_tick']['frac']})
        scheme = getattr(st, 'halo_mode', 'ghost')
        out = {
            'metric': 'million cell-steps/sec (grid_cells x timesteps / wall_s), %s %dx%d over %d GPUs' % (
                {'fenton': '4v', 'br': 'BR', 'court': 'Courtemanche'}[args.model], H, m.width, world),
            'value': round(value, 1), 'unit': 'Mcell-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(wall * 1000.0 / args.steps, 6), 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s %dx%d grid in %d row blocks of %d rows x %d cols%s, dt=0.1 ms, phase-field hole, '
                                   'S1 + S2; 1 step = 1 tick = %d sub-steps'
                                   % (args.model, H, m.width, world, H // world, m.width,
                                      ' (BASELINE configs[3])' if cfgi == 3 else
                                      (' (%s scaling of BASELINE configs[%d]\'s model by rows)' % ('strong' if strong else 'weak',
                                                                                           {'fenton': 1, 'br': 2, 'court': 4}[args.model])), spt),
                       'sub_steps_per_tick': spt, 'fused_sub_steps_per_launch': fused, 'launches_per_tick': per_tick,
                       'arithmetic': 'exact (one rounding per reference op)' if args.exact else 'fast_math (default policy)',
                       'parallelism': ('row-block x%d; one ghost row of the potential, exchanged after every sub-step (%d per tick), '
                                       'edge rows first, interior on a second stream' % (world, spt)) if scheme == 'rows1' else
                                      ('row-block x%d; ghost zone %d rows (= %d ticks): one point-to-point send/recv pair per '
                                       'neighbour every %d ticks, %d arrays in one contiguous message, interior '
                                       'overlapped on a second stream on tall blocks' % (world, ghost, halo_ticks, halo_ticks, halo_n)),
                       'halo_scheme': scheme,
                       'halo_transport': getattr(st, 'halo_path', 'none'), 'backend': backend,
                       'ranks_in_communicator': int(nranks.item()),
                       'halo_wait_s_max_rank': round(r['halo_wait'], 4), 'setup_ticks': r['setup_ticks'],
                       'timing': 'median of %d regions of %d ticks; per region the max over ranks' % (args.repeats, args.steps)},
            'repeats': args.repeats, 'wall_ms_per_region': [round(w * 1e3, 4) for w in walls],
            'roofline': roof,
            'predicted': predicted_figure(args.size, world, scheme, 'library' if getattr(st, 'rccl_direct', False) else 'torch')
                         if args.model == 'fenton' and strong and not args.exact else None,
        }

    # ---- side legs: everything the first multi-GPU run has to answer beyond the headline --------------------------------------
    # The headline is complete here.  It is written out (stderr, and the file the parent of self-spawned ranks watches) BEFORE
    # any side leg starts, every leg is bounded, and a watchdog holds the whole lot to --side-budget seconds: when it runs out —
    # a rank stuck in a collective cannot be called back — rank 0 prints the line as it stands and every rank leaves, so a leg
    # that fails or stalls costs its own key (and the legs behind it), never the headline.
    legs_done, legs_state = {}, {'current': None}
    headline_file = os.environ.get('FIBTF_HEADLINE_FILE')

    def emit_early():
        if out is None:
            return
        early = dict(out, side_legs_note='written before the side legs started')
        print('bench.py: headline before the side legs: %s' % json.dumps(early), file=sys.stderr, flush=True)
        if headline_file:
            with open(headline_file + '.tmp', 'w') as f:
                f.write(json.dumps(early))
            os.replace(headline_file + '.tmp', headline_file)

    def give_up():
        # (the watchdog's thread) nothing that needs the other ranks or the device from here on
        if out is not None:
            out.update(legs_done)
            out['side_legs_timed_out'] = 'the side legs ran out of their %d s while in %r; the legs finished until then are in ' \
                                         'this line, the headline above was complete before they started' % (args.side_budget, legs_state['current'])
            print(json.dumps(out), flush=True)
        sys.stderr.flush()
        os._exit(0)

    side = world > 1 and not args.no_side_legs
    watchdog = None
    if side:
        emit_early()
        watchdog = threading.Timer(args.side_budget, give_up)
        watchdog.daemon = True
        watchdog.start()

    def leg(name, fn):
        legs_state['current'] = name
        try:
            res = fn()
        except Exception as e:                          # (a failure on ONE rank leaves the others in a collective: the watchdog)
            res = {'error': '%s: %s' % (type(e).__name__, e)}
            print('bench.py: side leg %s failed on rank %d: %s' % (name, rank, res['error']), file=sys.stderr, flush=True)
        if rank == 0 and res is not None:
            legs_done[name] = res

    parity = {}
    if side:
        # (a) the N-rank state against the one-device state of the same ticks, bitwise: from where the timed regions
        # left it, two exchange cycles further; rank 0 repeats those ticks on ONE handle after the group is gone
        def gather_parity():
            T = 2 * halo_ticks
            before = st.get_state(-1)
            st.step(T)
            after = st.get_state(-1)
            if rank == 0:
                parity.update(T=T, before=before, after=after)
            return None
        leg('sharded_equals_single', gather_parity)

    st.sync()
    dist.barrier()
    direct = getattr(st, 'rccl_direct', False)
    st.close()                                          # (releases the library's communicator, if any, first)
    dist.barrier()

    if side:
        base = copy.copy(args)
        base.scaling = 'strong'
        # (b) north_star's own series: the 512x512 grid over the same N ranks
        if not (args.size == 512 and strong and getattr(args, 'halo', None) in (None, 'ghost')) and 512 // world >= 10:
            a = copy.copy(base)
            a.size, a.halo, a.halo_ticks = 512, None, 0
            leg('north_star_512', lambda: side_leg(a, 512, local, rank, world, 200, 3, 40,
                                                   "north_star: Fenton 4v 512x512 over %d GPUs (strong scaling of BASELINE configs[1])" % world))
        # (c) north_star's literal halo scheme on the headline's grid: one ghost row of the potential per sub-step
        if getattr(args, 'halo', None) != 'rows1' and not args.skip:
            a = copy.copy(base)
            a.size, a.halo, a.halo_ticks = args.size, 'rows1', 0
            leg('rows1_leg', lambda: side_leg(a, H, local, rank, world, 20, 3, 4,
                                              "%s %dx%d over %d GPUs, halo scheme 'rows1' (one ghost row of the potential, exchanged after "
                                              "every sub-step, interior on a second stream)" % (args.model, H, args.size, world)))
        # (d) LAST, because it is the one path no box has exercised between two devices: the library-issued exchange (grouped
        # ncclSend/ncclRecv on the compute stream, FIBTF_HALO=direct) on the headline's grid — one MI355X talking to itself ran a
        # configs[3] block 10 % faster through it than through torch's batch_isend_irecv (profiles/r04_exchange_tick_cost.txt).
        # Its state after the timed ticks is compared bit for bit with a fresh run of the same ticks through the default transport.
        if os.environ.get('FIBTF_HALO', 'torch') != 'direct' and getattr(args, 'halo', None) != 'rows1':
            def library_leg():
                a = copy.copy(base)
                a.size, a.halo, a.halo_ticks = args.size, getattr(args, 'halo', None), args.halo_ticks
                os.environ['FIBTF_HALO'] = 'direct'
                try:
                    r2 = sharded_run(a, H, local, args.steps, 3, 2 * halo_ticks, min(args.warmup, 8))
                finally:
                    os.environ.pop('FIBTF_HALO', None)
                m2, st2 = r2['m'], r2['st']
                nticks = r2['setup_ticks'] + min(args.warmup, 8) + 3 * args.steps    # what sharded_run has advanced
                got = st2.get_state(-1)
                used = getattr(st2, 'halo_path', 'none')
                st2.sync()
                st2.close()
                m3, (loc3, amp3, s2_ms3) = make_model(a, height=H, device=local)       # the same ticks (and S2) through the default transport
                m3.define()
                m3.add_pace_op('s2', loc3, amp3)
                driver_loop(m3, m3._stepper, m3.millisecond_to_step(s2_ms3), a.model == 'court')(nticks)
                want = m3._stepper.get_state(-1)
                m3._stepper.sync()
                m3._stepper.close()
                if rank != 0:
                    return None
                return {'config': 'the headline\'s grid with the halo exchange issued by the library itself', 'value': round(r2['value'], 1),
                        'unit': 'Mcell-steps/s', 'n_gpus': world, 'ms_per_step': round(r2['wall'] * 1000.0 / args.steps, 6),
                        'steps': args.steps, 'wall_ms_per_region': [round(w * 1e3, 4) for w in r2['walls']],
                        'halo_transport': used, 'halo_wait_s_max_rank': round(r2['halo_wait'], 4),
                        'equals_default_transport_bitwise': bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))),
                        'ticks_compared': nticks,
                        'predicted': predicted_figure(args.size, world, 'ghost', 'library') if args.model == 'fenton' and not args.exact else None}
            leg('library_transport_leg', library_leg)
        legs_state['current'] = 'teardown'

    # orderly teardown: every rank has drained its stream and released the library's communicator before anyone leaves
    dist.barrier()
    dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if out is not None:
        out.update(legs_done)
    if out is not None and (not args.no_single_leg or parity):
        # rank 0, after the group is gone (the other ranks have left): the SAME grid on this one device — (1) the ticks of the
        # parity leg repeated on one handle, compared bit for bit; (2) timed like the N = 1 line times its workload: the
        # denominator a scaling figure of this series needs (the N = 1 line of bench.py is BASELINE configs[1], another grid)
        try:
            a = copy.copy(args)
            a.steps, a.setup, a.warmup, a.repeats = max(10, min(args.steps, 100)), 30, 10, 3
            if parity:
                try:
                    m1, _ = make_model(a, height=H, device=local)
                    m1.define()
                    s1 = m1._stepper
                    s1.set_state(-1, parity['before'])
                    s1.step(parity['T'])
                    got = s1.get_state(-1)
                    plan1 = plan_text(s1)
                    s1.close()
                    same = bool(np.array_equal(got.view(np.uint32), parity['after'].view(np.uint32)))
                    out['sharded_equals_single'] = {
                        'equal_bitwise': same, 'max_abs_diff': float(np.abs(got - parity['after']).max()), 'ticks': parity['T'],
                        'arrays': int(got.shape[0]), 'note': 'state gathered from the %d ranks after the timed regions, %d more ticks '
                        '(two exchange cycles) on the ranks, the same %d ticks from the same state on ONE handle on rank 0\'s device '
                        '(its own launch plan: %s); every array compared bit for bit' % (world, parity['T'], parity['T'], plan1)}
                except Exception as e:
                    out['sharded_equals_single'] = {'error': '%s: %s' % (type(e).__name__, e)}
                parity.clear()
            if not args.no_single_leg:
                v1, ms1, roof1, _, _, m1 = measure_single(a, args.exact, False, snapshots=False, height=H, device=local)
                out['single_device_same_grid'] = {
                    'value': round(v1, 1), 'unit': 'Mcell-steps/s', 'ms_per_step': round(ms1, 6), 'steps': a.steps,
                    'plan': plan_text(m1._stepper), 'roofline_frac': roof1['frac'],
                    'note': 'the whole %dx%d grid as ONE handle on rank 0\'s device, median of 3 regions of %d ticks, measured in '
                            'this run after the ranks had finished' % (H, m1.width, a.steps)}
                m1._stepper.close()
                pred = out.get('predicted') or {}
                pred1 = predicted_figure(args.size, 1, 'ghost', 'torch') if pred else None
                out['scaling_efficiency'] = {
                    'speedup_vs_single_device_same_grid': round(out['value'] / v1, 4),
                    'efficiency': round(out['value'] / v1 / world, 4), 'n_gpus': world,
                    'predicted_speedup': round(pred['value'] / pred1['value'], 4) if pred and pred1 else None,
                    'predicted_efficiency': round(pred['value'] / pred1['value'] / world, 4) if pred and pred1 else None,
                    'note': 'value / single_device_same_grid.value (the same grid, the same run, rank 0\'s device) and that over the '
                            'number of GPUs; the prediction is profiles/r03_predicted_scaling.txt\'s (made on one MI355X before any '
                            'multi-GPU run)'}
        except Exception as e:                          # never lose the result line over the side leg
            out['single_device_same_grid'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if out is not None and not args.no_cpu:             # rank 0, after the group is gone: the other ranks have left
        out['cpu_baseline'] = cpu_baseline(args, height=H, seconds=8.0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000, help='ticks timed per region (1 tick = dt_per_step sub-steps)')
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps ticks; the median is reported')
    ap.add_argument('--setup', type=int, default=4000, help='untimed set-up ticks before the warm-up: 50 ms at 512x512, the time the '
                    'shader clock takes to settle after idling (400 ticks left a tick 3-4 %% slower: 12.3 vs 11.95 us)')
    ap.add_argument('--model', default='fenton', choices=['fenton', 'br', 'court'])
    ap.add_argument('--size', type=int, default=0, help='grid width (and height unless --scaling weak); default: 512 at '
                                                        'N=1 (1024 court), 4096 at N>1 (512 with --scaling weak)')
    ap.add_argument('--rows-per-gpu', type=int, default=512)
    ap.add_argument('--scaling', default=None, choices=['weak', 'strong'],
                    help='N>1: strong (default) = the fixed size x size grid split over the N GPUs (BASELINE configs[3] at '
                         'the default size 4096; --size 512 = north_star\'s "512x512 at 1/2/4/8"); weak = rows-per-gpu '
                         'rows on every GPU')
    ap.add_argument('--exact', action='store_true', help="config['fast_math']=False: one rounding per reference op")
    ap.add_argument('--no-exact-leg', action='store_true', help='N=1: skip the second, rounding-faithful measurement')
    ap.add_argument('--no-cheby', action='store_true')
    ap.add_argument('--skip', action='store_true')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-single-leg', action='store_true', help='N>1: skip the one-device run of the same grid on rank 0')
    ap.add_argument('--no-config-legs', action='store_true', help="N=1: skip the `configs` legs (BASELINE's other single-GPU configurations)")
    ap.add_argument('--halo-ticks', type=int, default=0, help='N > 1: ticks between two halo exchanges (ghost zone depth); 0 = '
                    'chosen from the block height (at most 4, extra rows within half of the block)')
    ap.add_argument('--halo', default=None, choices=['ghost', 'rows1'],
                    help="N > 1: 'ghost' (default) = multi-tick ghost zone of all arrays; 'rows1' = north_star's literal scheme, one "
                         "ghost row of the potential exchanged after every sub-step, edge rows first, interior on a second stream")
    ap.add_argument('--no-side-legs', action='store_true', help='N > 1: the headline only — no on-hardware parity check against one '
                    'device, no 512x512 leg, no rows1 leg')
    ap.add_argument('--side-budget', type=int, default=420, help='N > 1: seconds all side legs together may take before rank 0 prints '
                    'the line as it stands and every rank leaves')
    ap.add_argument('--spawn-timeout', type=int, default=1500, help='N > 1 without a launcher: seconds before the ranks are stopped')
    ap.add_argument('--force-dist', action='store_true', help='run the rank path even in a one-rank group (rehearsal)')
    args = ap.parse_args()
    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus < 1:
        ap.error('--gpus must be at least 1')
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args))
    world = int(env_world or '1')
    if world != args.gpus:
        print('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree' % (args.gpus, world),
              file=sys.stderr)
        sys.exit(2)
    multi = world > 1 or args.force_dist
    if args.scaling is None:
        args.scaling = 'strong'
    if not args.size:
        args.size = (4096 if args.scaling == 'strong' else 512) if world > 1 else (1024 if args.model == 'court' else 512)
    out = bench_ranks(args) if multi else bench_single(args)
    if out is None:
        return
    print(json.dumps(out))


if __name__ == '__main__':
    main()
