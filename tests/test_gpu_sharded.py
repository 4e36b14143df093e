"""-m gpu: the row-block driver with the REAL HIP engine.  The GPU box has one MI355X, so two (and
three) ranks share device 0 and exchange halos through gloo with host staging (RCCL refuses two
ranks on one device); everything else is the production path: torch-owned slabs handed to fibhip as
ext_slab, ghost rows, row_offset-aware kernels, step_edges / step_interior on two streams /
step_commit.  The gathered result must be BITWISE equal to the single-handle run on the same GPU."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_sharded_cpu import launch  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [
    (2, {'model': 'fenton', 'H': 128, 'W': 96, 'diff': 1.5, 'hole': (40, 64, 9), 'ticks': 12, 's2': 5, 'amp': 1.0}),
    (3, {'model': 'fenton', 'H': 131, 'W': 70, 'diff': 1.1, 'hole': (30, 50, 7), 'ticks': 7, 's2': 2, 'amp': 1.0}),
    (2, {'model': 'br', 'H': 90, 'W': 77, 'diff': 0.809, 'hole': (30, 40, 8), 'ticks': 9, 's2': 4, 'amp': 10.0,
         'cheby': True, 'skip': False}),
    (3, {'model': 'br', 'H': 100, 'W': 64, 'diff': 0.809, 'hole': (30, 40, 8), 'ticks': 6, 's2': 3, 'amp': 10.0,
         'cheby': False, 'skip': True}),
    (2, {'model': 'court', 'H': 70, 'W': 66, 'diff': 0.809, 'hole': (30, 30, 6), 'ticks': 25, 's2': 12, 'amp': 10.0}),
    (2, {'model': 'fenton', 'H': 128, 'W': 96, 'diff': 1.5, 'hole': (40, 64, 9), 'ticks': 9, 's2': 5, 'amp': 1.0,
         'halo_ticks': 1}),
    (2, {'model': 'fenton', 'H': 300, 'W': 96, 'diff': 1.5, 'hole': (40, 150, 9), 'ticks': 11, 's2': 5, 'amp': 1.0,
         'halo_ticks': 4}),
    (3, {'model': 'br', 'H': 120, 'W': 64, 'diff': 0.809, 'hole': (30, 40, 8), 'ticks': 7, 's2': 3, 'amp': 10.0,
         'cheby': True, 'skip': True, 'halo_ticks': 2}),
    (2, {'model': 'court', 'H': 70, 'W': 66, 'diff': 0.809, 'hole': (30, 30, 6), 'ticks': 23, 's2': 12, 'amp': 10.0,
         'halo_ticks': 1}),
    # Courtemanche with a ghost zone 4 and 6 ticks deep: the ticks between two exchanges run as ONE temporally blocked launch
    # (3 ticks; 3 + 2) on the five aggregates of the slow variables, the tick that ends the cycle on its own
    (2, {'model': 'court', 'H': 90, 'W': 66, 'diff': 0.809, 'hole': (30, 40, 6), 'ticks': 27, 's2': 12, 'amp': 10.0,
         'halo_ticks': 4}),
    (3, {'model': 'court', 'H': 120, 'W': 70, 'diff': 0.809, 'hole': (30, 50, 6), 'ticks': 31, 's2': 14, 'amp': 10.0,
         'halo_ticks': 6}),
    # north_star's literal halo scheme ('rows1'): one ghost row of the potential exchanged after every sub-step, one
    # sub-step per launch, edge rows first and the interior on the second stream (with `split`)
    (2, {'model': 'fenton', 'H': 128, 'W': 96, 'diff': 1.5, 'hole': (40, 64, 9), 'ticks': 4, 's2': 2, 'amp': 1.0,
         'halo': 'rows1'}),
    (3, {'model': 'br', 'H': 100, 'W': 64, 'diff': 0.809, 'hole': (30, 40, 8), 'ticks': 5, 's2': 2, 'amp': 10.0,
         'cheby': True, 'skip': False, 'halo': 'rows1'}),
    (2, {'model': 'court', 'H': 70, 'W': 66, 'diff': 0.809, 'hole': (30, 30, 6), 'ticks': 23, 's2': 12, 'amp': 10.0,
         'halo': 'rows1'}),
    # four ranks on the one GPU (the box allows six processes on the card; the test runner itself holds it too)
    (4, {'model': 'fenton', 'H': 230, 'W': 70, 'diff': 1.5, 'hole': (30, 110, 9), 'ticks': 9, 's2': 4, 'amp': 1.0}),
    (4, {'model': 'br', 'H': 160, 'W': 64, 'diff': 0.809, 'hole': (30, 80, 8), 'ticks': 7, 's2': 3, 'amp': 10.0,
         'cheby': True, 'skip': False, 'halo_ticks': 2}),
    # the width of BASELINE configs[3] (4096 columns: 76-93 tiles per tile row, the edge-strip / interior split really
    # splits), four ranks, the default 4-tick ghost zone
    (4, {'model': 'fenton', 'H': 212, 'W': 4096, 'diff': 1.5, 'hole': (2048, 106, 30), 'ticks': 6, 's2': 2, 'amp': 1.0,
         'halo_ticks': 4}),
    # ONE array rewritten through set_state in the middle of an exchange cycle
    (3, {'model': 'fenton', 'H': 150, 'W': 96, 'diff': 1.5, 'hole': (40, 70, 9), 'ticks': 11, 's2': 8, 'amp': 1.0,
         'halo_ticks': 4, 'poke': (5, 0)}),
    # traced model files (tests/models/): the generated library behind the same row-block driver
    (2, {'model': 'ap', 'H': 128, 'W': 80, 'diff': 1.0, 'hole': (30, 60, 8), 'ticks': 11, 's2': 4, 'amp': 1.0}),
    (3, {'model': 'gated', 'H': 90, 'W': 64, 'diff': 2.0, 'hole': (30, 40, 6), 'ticks': 31, 's2': 14, 'amp': 10.0,
         'halo_ticks': 2}),
    (2, {'model': 'mrfhn', 'H': 100, 'W': 70, 'diff': 1.0, 'hole': (30, 50, 7), 'ticks': 9, 's2': 3, 'amp': 1.5}),
]


def single(case):
    from fib_tf_amd.fenton import Fenton4v
    from fib_tf_amd.br import BeelerReuter
    from fib_tf_amd.court import Courtemanche
    cfg = {'height': case['H'], 'width': case['W'], 'dt': 0.1, 'dt_per_plot': 10, 'diff': case['diff'],
           'duration': 1000, 'cheby': case.get('cheby', False), 'skip': case.get('skip', False)}
    if case['model'] in ('ap', 'ms', 'gated', 'mrfhn'):
        from traced_cases import make_model
        m = make_model(case['model'], case['H'], case['W'], case['hole'])
    else:
        m = {'fenton': Fenton4v, 'br': BeelerReuter, 'court': Courtemanche}[case['model']](cfg)
        m.add_hole_to_phase_field(*case['hole'])
    m.define()
    m.add_pace_op('s2', 'luq', case['amp'])
    m.duration = case['ticks'] * m.dt_per_step * m.dt + 1e-9
    trend = []
    for i in m.run():
        if case.get('poke') and i == case['poke'][0]:
            v = case['poke'][1]
            m._stepper.set_state(v, m._State[m.VAR_NAMES[v]].eval() * np.float32(0.5) + np.float32(0.125))
        if case['model'] in ('court', 'gated') and i % 10 == 0:
            m.fire_op('slow')
            m.fire_op('trend')
            trend.append(m._Trend.eval())
        if i == case['s2']:
            m.fire_op('s2')
    return np.stack([m._State[n].eval() for n in m.VAR_NAMES]), np.array(trend, np.float32)


@pytest.mark.parametrize('world,case', CASES,
                         ids=['%s-x%d-h%s-%dx%d%s' % (c['model'], w, c.get('halo_ticks', 'd'), c['H'], c['W'],
                                                      '-' + c['halo'] if c.get('halo') else '') for w, c in CASES])
@pytest.mark.parametrize('split', ['auto', 'split'])
def test_sharded_hip_equals_single_handle(gpu_lib, world, case, tmp_path, split, monkeypatch):
    """split: force the two-stream edge-strips / interior launch on every exchange tick (the library only
    chooses it by itself for blocks that are several CU rounds tall)"""
    if split == 'split':
        monkeypatch.setenv('FIBHIP_SPLIT', '1')
    else:
        monkeypatch.delenv('FIBHIP_SPLIT', raising=False)
    want, trend = single(case)
    out = launch(world, dict(case, engine='hip'), tmp_path)
    assert np.array_equal(out['full'], want), 'max|d| = %g' % np.abs(out['full'] - want).max()
    if case['model'] in ('court', 'gated'):
        assert np.array_equal(out['trend'], trend)


def test_rccl_self_exchange_on_slab_views(gpu_lib, tmp_path):
    """the RCCL leg itself, as far as one GPU can run it: backend nccl, one rank, the production P2POp pattern on the
    engine's slab views and stream, sent to itself (tests/rccl_self_worker.py)"""
    import subprocess
    from test_sharded_cpu import free_port
    out = str(tmp_path / 'ok.npy')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_self_worker.py'),
                        str(free_port()), out], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    ok = np.load(out)
    assert ok.all(), ok


def _bench(args, env_extra, timeout=600):
    import json
    import subprocess
    from test_sharded_cpu import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FIBTF_DIST_BACKEND', 'FIBTF_ONE_DEVICE')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_rank_path_on_rccl_backend_single_rank(gpu_lib):
    """bench.py's rank path on the real backend with a one-rank group (--force-dist): process-group init with
    device_id, barriers, the MAX all-reduce of the timing, the result line (what two RCCL ranks would add is only the
    halo messages themselves, covered by the self-exchange test above)"""
    r, line = _bench(['--gpus', '1', '--force-dist', '--steps', '100', '--warmup', '10', '--setup', '20', '--no-cpu'],
                     {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 1 and line['value'] > 1000 and line['config']['backend'] == 'nccl'
    assert line['config']['ranks_in_communicator'] == 1 and len(line['wall_ms_per_region']) == 3


def test_bench_more_gpus_than_devices_fails_loudly(gpu_lib):
    """`python bench.py --gpus 2` on the one-GPU box: no result line, a non-zero exit and the reason"""
    r, line = _bench(['--gpus', '2', '--steps', '10', '--warmup', '2'], {})
    assert r.returncode != 0 and line is None
    assert 'needs 2 HIP device' in r.stderr


def test_bench_spawns_its_ranks(gpu_lib):
    """`python bench.py --gpus 2` without a launcher starts two child ranks and relays rank 0's line.  The box has one
    GPU, so the ranks share it and talk through gloo (FIBTF_ONE_DEVICE / FIBTF_DIST_BACKEND: rehearsal switches);
    spawning, rendezvous, the timed regions, the kernel probe and the teardown are the production code."""
    r, line = _bench(['--gpus', '2', '--size', '512', '--steps', '40', '--warmup', '8', '--setup', '16', '--no-cpu'],
                     {'FIBTF_ONE_DEVICE': '1', 'FIBTF_DIST_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['ranks_in_communicator'] == 2
    assert line['value'] > 1000 and line['roofline']['us_per_launch'] > 0 and line['roofline']['whole_tick']['us_per_tick'] > 0
    # the same grid on one device, measured by rank 0 after the ranks have left: what a scaling figure divides by
    assert line['single_device_same_grid']['value'] > 1000, line['single_device_same_grid']
    # side legs (bench.py `bench_ranks`): the ranks' state against ONE handle advancing the same ticks, bit for bit, on the
    # hardware the line was measured on; north_star's literal halo scheme on the same ranks; the scaling figure
    eq = line['sharded_equals_single']
    assert eq.get('equal_bitwise') is True and eq['max_abs_diff'] == 0.0 and eq['ticks'] >= 2, eq
    assert line['rows1_leg']['value'] > 100 and line['rows1_leg']['halo_scheme'] == 'rows1', line['rows1_leg']
    assert 'north_star_512' not in line                    # (the headline already is that grid)
    se = line['scaling_efficiency']
    assert abs(se['efficiency'] * 2 - se['speedup_vs_single_device_same_grid']) < 1e-3
    assert line['predicted']['value'] > 0 and 'side_legs_timed_out' not in line


def test_bench_side_legs_on_the_headline_grid(gpu_lib):
    """the driver's plain `bench.py --gpus N` (BASELINE configs[3]: 4096x4096) rehearsed with 2 ranks sharing the one GPU:
    the line carries the on-hardware parity statement, north_star's 512x512 series on the same ranks and the rows1 leg,
    each with the figure predicted for it"""
    r, line = _bench(['--gpus', '2', '--steps', '8', '--warmup', '4', '--setup', '8', '--no-cpu'],
                     {'FIBTF_ONE_DEVICE': '1', 'FIBTF_DIST_BACKEND': 'gloo'}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line['n_gpus'] == 2 and '4096x4096' in line['metric'] and 'configs[3]' in line['config']['workload']
    assert line['sharded_equals_single'].get('equal_bitwise') is True, line['sharded_equals_single']
    ns = line['north_star_512']
    assert ns['value'] > 1000 and ns['n_gpus'] == 2 and ns['predicted']['value'] > 0 and ns['rank0_kernels_us_per_tick'] > 0, ns
    assert line['rows1_leg']['value'] > 100 and line['rows1_leg']['predicted']['value'] > 0, line['rows1_leg']
    # (gloo ranks cannot take the library's own RCCL exchange: the leg runs on the staged transport and says so; its bitwise
    # comparison with a fresh run of the same ticks holds all the same)
    lib = line['library_transport_leg']
    assert lib['value'] > 100 and lib['equals_default_transport_bitwise'] is True and 'host-staged' in lib['halo_transport'], lib
    assert line['single_device_same_grid']['value'] > 1000 and line['scaling_efficiency']['predicted_efficiency'] > 0
    # the headline was written out before the side legs started
    assert 'headline before the side legs' in r.stderr


def test_bench_under_the_drivers_launcher(gpu_lib):
    """the driver's own way of starting N ranks — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — rehearsed with 2 gloo ranks sharing the one GPU: one JSON
    line on stdout (rank 0's), side legs included, exit code 0 (no rank leaves early, nothing for the elastic agent to flag)"""
    import json
    import subprocess
    from test_sharded_cpu import free_port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', FIBTF_ONE_DEVICE='1', FIBTF_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '8', '--warmup', '4',
           '--setup', '8', '--no-cpu']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 8 and line['warmup'] == 4 and line['value'] > 1000
    assert line['sharded_equals_single'].get('equal_bitwise') is True, line['sharded_equals_single']
    for key in ('north_star_512', 'rows1_leg', 'library_transport_leg', 'single_device_same_grid', 'scaling_efficiency'):
        assert key in line, key


def test_direct_rccl_exchange_self(gpu_lib, tmp_path):
    """the opt-in direct halo path (FIBTF_HALO=direct): RCCL bound with dlopen, communicator created by the library,
    one grouped ncclSend/ncclRecv exchange on the handle's stream — with a one-rank communicator (tests/rccl_direct_worker.py)"""
    import subprocess
    out = str(tmp_path / 'ok.npy')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_direct_worker.py'), out],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.load(out).all()


@pytest.mark.parametrize('world,case', [c for c in CASES if c[1]['model'] in ('fenton', 'br', 'ap')][:6],
                         ids=lambda v: str(v) if isinstance(v, int) else v['model'] + str(v['H']))
def test_library_halo_messages_equal_single_handle(gpu_lib, world, case, tmp_path, monkeypatch):
    """FIBTF_HALO=plan: the exchange executes the message list the C library builds for its own RCCL transport
    (fibhip_halo_plan: which rows of the interleaved slab go to / come from which rank) through host staging —
    with 2-4 ranks on the one GPU the result must equal the single-handle run bit for bit"""
    monkeypatch.setenv('FIBTF_HALO', 'plan')
    monkeypatch.delenv('FIBHIP_SPLIT', raising=False)
    want, _ = single(case)
    out = launch(world, dict(case, engine='hip'), tmp_path)
    assert np.array_equal(out['full'], want), 'max|d| = %g' % np.abs(out['full'] - want).max()
    assert int(out['plan_exchanges']) >= 1                 # the library's message list really was the transport
