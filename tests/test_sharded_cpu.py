"""not-gpu: the multi-rank path (fib_tf_amd/sharded.py) with world_size 2 and 3 over gloo on CPU.
The product's own host logic runs — row slicing, ghost rows, batched isend/irecv halo exchange,
gather for image()/eval(), pacing in global coordinates, the model classes' run() loop — with the
CPU test engine (tests/cpu_engine.py, oracle arithmetic) in place of the HIP engine.
Expectation: the gathered N-rank result is BITWISE equal to the single-domain oracle run."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(world, case, tmp_path):
    from sharded_worker import run_case
    ctx = mp.get_context('spawn')
    port = free_port()
    procs = [ctx.Process(target=run_case, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail('rank hung')
        assert p.exitcode == 0
    return np.load(os.path.join(str(tmp_path), 'out.npz'))


def single_domain(case, orc):
    from fib_tf_amd.ionic import IonicModel
    H, W = case['H'], case['W']
    g = IonicModel({'height': H, 'width': W})
    g.add_hole_to_phase_field(*case['hole'])
    rect = g.pace_rect('luq')
    if case['model'] == 'fenton':
        s = np.zeros((4, H, W), np.float32)
        s[1:3] = 1.0
        s[0][:, 1] = 1.0
        for i in range(case['ticks']):
            orc.fenton_run(s, 0.1, case['diff'], g.phase, 10)
            if case.get('poke') and i == case['poke'][0]:
                v = case['poke'][1]
                s[v] = s[v] * np.float32(0.5) + np.float32(0.125)
            if i == case['s2']:
                s[0] = orc.pace(s[0], *rect, case['amp'], 0.0)
        return s, None
    if case['model'] == 'br':
        from fib_tf_amd.br import BeelerReuter
        s = np.empty((8, H, W), np.float32)
        for i, v in enumerate((-84.624, 1e-4, 0.01, 0.988, 0.975, 0.003, 0.994, 0.0001)):
            s[i] = v
        s[0][:, 1] = 10.0
        tbl = BeelerReuter({'height': 8, 'width': 8}).chebyshev_table().astype(np.float32) if case['cheby'] else None
        for i in range(case['ticks']):
            orc.br_run(s, 0.1, case['diff'], g.phase, tbl, case['skip'], 1)
            if i == case['s2']:
                s[0] = orc.pace(s[0], *rect, case['amp'], -90.0)
        return s, None
    from fib_tf_amd.court import INITIAL
    s = np.empty((21, H, W), np.float32)
    for i, (_, v) in enumerate(INITIAL):
        s[i] = v
    s[0][:, :25] = 20.0
    trend = []
    for i in range(case['ticks']):
        orc.court_run(s, 0.1, case['diff'], g.phase, True, i, 1)
        if i % 10 == 0:
            trend.append([s[0][W // 2, 20], s[1][W // 2, 20]])
        if i == case['s2']:
            s[0] = orc.pace(s[0], *rect, case['amp'], -100.0)
    return s, np.array(trend, np.float32)


CASES = [
    (2, {'model': 'fenton', 'H': 48, 'W': 40, 'diff': 1.5, 'hole': (20, 24, 5), 'ticks': 6, 's2': 3, 'amp': 1.0}),
    (3, {'model': 'fenton', 'H': 50, 'W': 37, 'diff': 1.1, 'hole': (18, 30, 4), 'ticks': 4, 's2': 1, 'amp': 1.0}),
    (2, {'model': 'br', 'H': 30, 'W': 44, 'diff': 0.809, 'hole': (20, 12, 4), 'ticks': 8, 's2': 4, 'amp': 10.0,
         'cheby': True, 'skip': False}),
    (3, {'model': 'br', 'H': 33, 'W': 25, 'diff': 0.809, 'hole': (10, 15, 4), 'ticks': 6, 's2': 2, 'amp': 10.0,
         'cheby': False, 'skip': True}),
    (2, {'model': 'court', 'H': 60, 'W': 56, 'diff': 0.809, 'hole': (28, 30, 5), 'ticks': 25, 's2': 12, 'amp': 10.0}),
    # one exchange per tick (halo_ticks=1) and a ghost zone 3 ticks deep with a tick count that is not a multiple
    (2, {'model': 'fenton', 'H': 80, 'W': 40, 'diff': 1.5, 'hole': (20, 40, 5), 'ticks': 5, 's2': 2, 'amp': 1.0,
         'halo_ticks': 1}),
    (2, {'model': 'fenton', 'H': 80, 'W': 40, 'diff': 1.5, 'hole': (20, 40, 5), 'ticks': 7, 's2': 4, 'amp': 1.0,
         'halo_ticks': 3}),
    (3, {'model': 'court', 'H': 45, 'W': 56, 'diff': 0.809, 'hole': (28, 20, 5), 'ticks': 23, 's2': 11, 'amp': 10.0,
         'halo_ticks': 1}),
    # more ranks: interior ranks with two neighbours on both sides, uneven blocks
    (4, {'model': 'fenton', 'H': 103, 'W': 33, 'diff': 1.5, 'hole': (16, 50, 6), 'ticks': 7, 's2': 3, 'amp': 1.0,
         'halo_ticks': 2}),
    (5, {'model': 'br', 'H': 73, 'W': 21, 'diff': 0.809, 'hole': (10, 36, 4), 'ticks': 6, 's2': 2, 'amp': 10.0,
         'cheby': True, 'skip': False, 'halo_ticks': 2}),
    # the 8-rank geometry of BASELINE configs[3] (4096 rows over 8 GPUs), scaled down: uneven blocks, six interior
    # ranks, an exchange every tick and the default 4-tick ghost zone (tick count not a multiple of the cycle)
    (8, {'model': 'fenton', 'H': 93, 'W': 24, 'diff': 1.5, 'hole': (12, 46, 5), 'ticks': 5, 's2': 2, 'amp': 1.0,
         'halo_ticks': 1}),
    (8, {'model': 'fenton', 'H': 331, 'W': 24, 'diff': 1.5, 'hole': (12, 160, 6), 'ticks': 10, 's2': 3, 'amp': 1.0,
         'halo_ticks': 4}),
    # north_star's literal halo scheme ('rows1'): ONE ghost row of the potential, exchanged after every sub-step
    (2, {'model': 'fenton', 'H': 48, 'W': 40, 'diff': 1.5, 'hole': (20, 24, 5), 'ticks': 4, 's2': 2, 'amp': 1.0,
         'halo': 'rows1'}),
    (8, {'model': 'fenton', 'H': 93, 'W': 24, 'diff': 1.5, 'hole': (12, 46, 5), 'ticks': 3, 's2': 1, 'amp': 1.0,
         'halo': 'rows1'}),
    (3, {'model': 'court', 'H': 45, 'W': 56, 'diff': 0.809, 'hole': (28, 20, 5), 'ticks': 23, 's2': 11, 'amp': 10.0,
         'halo': 'rows1'}),
    # the ghost depth chosen from the block height (halo_ticks 0): 331 rows over 8 ranks -> 41-row blocks -> 3 ticks
    (8, {'model': 'fenton', 'H': 331, 'W': 24, 'diff': 1.5, 'hole': (12, 160, 6), 'ticks': 8, 's2': 3, 'amp': 1.0,
         'halo_ticks': 0}),
    # ONE array rewritten through set_state in the middle of an exchange cycle (the other arrays' outer ghost rows
    # are stale at that point: the cycle position must survive the call)
    (3, {'model': 'fenton', 'H': 150, 'W': 28, 'diff': 1.5, 'hole': (14, 70, 5), 'ticks': 11, 's2': 8, 'amp': 1.0,
         'halo_ticks': 4, 'poke': (5, 0)}),
]


@pytest.mark.parametrize('world,case', CASES,
                         ids=['%s-x%d-h%s%s%s' % (c['model'], w, c.get('halo_ticks', 'd'), '-poke' if c.get('poke') else '',
                                                  '-' + c['halo'] if c.get('halo') else '') for w, c in CASES])
def test_sharded_equals_single_domain(world, case, tmp_path, orc):
    out = launch(world, case, tmp_path)
    want, trend = single_domain(case, orc)
    assert out['blocks'].shape == (world, 2) and int(out['blocks'][:, 1].sum()) == case['H']
    assert np.array_equal(out['full'], want), 'max|d| = %g' % np.abs(out['full'] - want).max()
    if trend is not None:
        assert np.array_equal(out['trend'], trend)
    if case.get('halo') == 'rows1':
        assert int(out['halo_ticks']) == 1
    if case.get('halo_ticks', 4) == 0:
        assert int(out['halo_ticks']) == 3          # 41-row blocks, 10 sub-steps per tick: 1 + 41 // 20


def test_bench_refuses_more_gpus_than_devices():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself; on a box with fewer devices (this
    container has none) it must fail loudly instead of printing a one-GPU line"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FIBTF_ONE_DEVICE')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert 'needs 2 HIP device' in r.stderr and '{' not in r.stdout
    # under a launcher, --gpus and WORLD_SIZE must agree
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], capture_output=True, text=True,
                       timeout=300, env=dict(env, WORLD_SIZE='1', RANK='0'))
    assert r.returncode != 0 and 'must agree' in r.stderr


def test_row_blocks():
    from fib_tf_amd.sharded import row_blocks
    assert row_blocks(512, 8) == [(64 * i, 64) for i in range(8)]
    assert row_blocks(10, 3) == [(0, 4), (4, 3), (7, 3)]
    b = row_blocks(4099, 8)
    assert sum(n for _, n in b) == 4099 and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(7))


def test_thin_blocks_rejected(tmp_path):
    """a halo deeper than a row block cannot be served by the nearest neighbour alone"""
    case = {'model': 'fenton', 'H': 16, 'W': 16, 'diff': 1.5, 'hole': (8, 8, 2), 'ticks': 1, 's2': 9, 'amp': 1.0}
    from sharded_worker import run_case
    ctx = mp.get_context('spawn')
    port = free_port()
    procs = [ctx.Process(target=run_case, args=(r, 2, port, case, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode not in (0, None) for p in procs)      # ValueError on every rank
