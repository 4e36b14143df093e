"""worker for tests/test_sharded_cpu.py: one gloo rank running the product's sharded driver
(fib_tf_amd/sharded.py + the model classes) over the CPU test engine"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def run_case(rank, world, port, case, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cpu_engine import OracleEngine
        from fib_tf_amd.fenton import Fenton4v
        from fib_tf_amd.br import BeelerReuter
        from fib_tf_amd.court import Courtemanche
        H, W, ticks = case['H'], case['W'], case['ticks']
        cfg = {'height': H, 'width': W, 'dt': 0.1, 'dt_per_plot': 10, 'diff': case['diff'], 'duration': 1000,
               'cheby': case.get('cheby', False), 'skip': case.get('skip', False),
               'halo_ticks': case.get('halo_ticks', 4), 'halo': case.get('halo')}
        if case.get('engine', 'oracle') == 'oracle':
            # CPU rehearsal: the product's sharded driver on the oracle-backed test engine.  The engine is a
            # constructor argument of ShardedStepper, not a configuration key of the models: the test puts it there.
            import fib_tf_amd.sharded as sharded

            class CpuShardedStepper(sharded.ShardedStepper):
                def __init__(self, *a, **kw):
                    kw['engine_factory'] = OracleEngine
                    kw.pop('library', None)              # (a table-specialised HIP library means nothing to it)
                    super().__init__(*a, **kw)
            sharded.ShardedStepper = CpuShardedStepper
            cfg['specialise'] = False                    # no device code is run: no per-table build either
        else:
            cfg['device'] = 0
        slow_trend = case['model'] in ('court', 'gated')
        if case['model'] in ('ap', 'ms', 'gated', 'mrfhn'):       # traced model files (tests/models/)
            from traced_cases import make_model
            extra = {k: cfg[k] for k in ('halo_ticks', 'halo', 'device') if k in cfg}
            m = make_model(case['model'], H, W, case['hole'], **extra)
        else:
            cls = {'fenton': Fenton4v, 'br': BeelerReuter, 'court': Courtemanche}[case['model']]
            m = cls(cfg)
            m.add_hole_to_phase_field(*case['hole'])
        m.define()
        m.add_pace_op('s2', 'luq', case['amp'])
        m.duration = ticks * m.dt_per_step * m.dt + 1e-9
        trend = []
        poke = case.get('poke')                          # (tick, var): rewrite ONE array in mid-cycle, on every rank
        for i in m.run():
            if poke and i == poke[0]:
                name = m.VAR_NAMES[poke[1]]
                m._stepper.set_state(poke[1], m._State[name].eval() * np.float32(0.5) + np.float32(0.125))
            if slow_trend and i % 10 == 0:
                m.fire_op('slow')
                m.fire_op('trend')
                trend.append(m._Trend.eval())
            if i == case['s2']:
                m.fire_op('s2')
        full = np.stack([m._State[n].eval() for n in m.VAR_NAMES])
        img = m.image()
        if rank == 0:
            np.savez(os.path.join(outdir, 'out.npz'), full=full, img=img, trend=np.array(trend, np.float32),
                     blocks=np.array(m._stepper.blocks), halo_ticks=m._stepper.halo_ticks,
                     plan_exchanges=getattr(m._stepper, 'plan_exchanges', 0))
    finally:
        dist.barrier()
        dist.destroy_process_group()
