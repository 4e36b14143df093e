"""which ingredient of bench.py's Beeler-Reuter flow makes the first stock-library launch die under rocprofv3? (one ingredient per run)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fib_tf_amd import _lib
from fib_tf_amd.br import BeelerReuter

mode = sys.argv[1]
m = BeelerReuter({'width': 512, 'height': 512, 'dt': 0.1, 'diff': 0.809, 'cheby': True, 'skip': False, 'dt_per_plot': 10,
                  'duration': 1000, 'fast_math': True})
if 'phase' in mode:
    m.add_hole_to_phase_field(150, 200, 40)
m.define()
st = m._stepper
st.step(50)
st.sync()
if 'image' in mode:
    m.image()
if 'timed' in mode:
    st.time_begin(); st.step(10); st.time_end()
if 'pace' in mode:
    m.add_pace_op('s2', 'luq', 10.0)
    m.fire_op('s2')
    st.sync()
if 'plan' in mode:
    print(st.launch_plan(), st.plan_tile(), st.ticks_per_launch())
if 'single' in mode:
    for _ in range(40):
        st.step(1)
    st.sync()
n = 1 << 30 if 'big' in mode else 1 << 26
if 'maps' in mode:
    for l in open('/proc/self/maps'):
        if 'r-xp' in l and ('.so' in l or 'python' in l):
            print(l.rstrip(), file=sys.stderr)
    sys.stderr.flush()
print(mode, 'copy bandwidth', _lib.copy_bandwidth(n, 2, 0))
