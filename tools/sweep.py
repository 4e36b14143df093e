#!/usr/bin/env python3
"""tuning sweep (GPU box): times every compiled kernel variant of a model on the bench workload.
usage: python tools/sweep.py [fenton|br|court] [size] [fast]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench

VARIANTS = {
    'fenton': ['10,44,25,-35', '5,54,21,-35', '10,44,25,-3', '10,44,28,-3', '10,44,32,-4', '10,44,36,-4', '10,44,40,-4', '10,44,44,-4',
               '5,54,21,-3', '5,54,23,-3', '5,54,22,-4', '5,54,27,-3', '5,54,32,-4', '5,54,40,-3', '5,54,44,-4', '5,54,56,-4',
               '2,60,18,-4', '10,32,32,512', '10,32,32,1024', '10,32,32,256', '5,32,32,256', '5,32,32,512', '5,32,16,256',
               '2,64,16,256', '2,32,32,256', '1,64,16,256', '1,64,4,256'],
    'br': ['2,60,19,-2', '2,60,19,-3', '3,58,19,-2', '5,54,21,-3', '5,54,21,-2', '5,32,32,256', '5,32,32,512', '1,64,16,256', '1,64,4,256'],
    'court': ['1,64,4,256', '1,64,8,256'],
}


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else 'fenton'
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    fast = len(sys.argv) > 3 and sys.argv[3] == 'fast'
    args = argparse.Namespace(model=model, size=size, exact=not fast, no_cheby=False, skip=False)
    for v in VARIANTS[model] + sys.argv[4:]:
        os.environ['FIBHIP_VARIANT'] = v
        m, _ = bench.make_model(args)
        m.define()
        st = m._stepper
        fused, per_tick = st.launch_plan()
        st.step(20)
        st.sync()
        n = 300 if size <= 1024 else 30
        best = 1e9
        for _ in range(3):
            ms, launches = st.time_steps(n)
            best = min(best, ms)
        cells = m.height * m.width
        mcs = cells * n * m.dt_per_step / (best * 1e-3) / 1e6
        print('%-8s %4d %-5s variant %-16s fused %2d launches/tick %2d  %8.2f us/tick %8.2f us/launch  %10.0f Mcell-steps/s'
              % (model, size, 'fast' if fast else 'exact', v, fused, per_tick, best * 1000 / n, best * 1000 / launches, mcs),
              flush=True)
        st.close()


if __name__ == '__main__':
    main()
