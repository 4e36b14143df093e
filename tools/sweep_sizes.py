"""plan selection check (GPU box): default plan against the candidate kernel shapes across grid sizes
(python tools/sweep_sizes.py > profiles/r02_sweep_sizes.txt)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'),'tools'))
import argparse, bench
def run(model, size, variant=None, k=None):
    for e in ('FIBHIP_VARIANT','FIBHIP_K'): os.environ.pop(e, None)
    if variant: os.environ['FIBHIP_VARIANT']=variant
    if k: os.environ['FIBHIP_K']=str(k)
    args = argparse.Namespace(model=model, size=size, exact=False, no_cheby=False, skip=False)
    m,_ = bench.make_model(args)
    if model=='br' and variant is None and k: pass
    m.define(); st=m._stepper
    st.step(20); st.sync()
    n = 200 if size <= 1024 else (60 if size <= 2048 else 20)
    best = min(st.time_steps(n)[0] for _ in range(3))
    print('%-7s %5d %-14s plan %-8s %8.2f us/tick %9.0f Mcs/s' % (model, size, variant or ('K=%s'%k if k else 'default'), st.launch_plan(), best*1000/n, m.height*m.width*n*m.dt_per_step/(best*1e-3)/1e6), flush=True)
    st.close()
FV = ('10,44,25,-3', '10,44,27,-3', '10,44,28,-3', '10,44,30,-3', '10,44,32,-4', '10,44,36,-4', '10,44,40,-4', '10,44,44,-4', '5,54,21,-3', '5,54,23,-3', '5,54,22,-4',
      '5,54,25,-3', '5,54,27,-3', '5,54,28,-3', '5,54,31,-3', '5,54,34,-3', '5,54,32,-4', '5,54,40,-3', '5,54,44,-4', '5,54,56,-4')
SIZES = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (384, 512, 576, 640, 704, 768, 832, 896, 960, 1024, 1280, 1536, 2048, 4096)
for size in SIZES:
    for v in (None,) + FV:
        run('fenton', size, v)
for size in ([] if len(sys.argv) > 1 else (512, 576, 640, 704, 768, 896, 1024)):
    for k in (None, 5, 1):
        run('br', size, None, k)

