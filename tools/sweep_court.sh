#!/bin/bash
# Courtemanche multi-tick launch shapes (GPU box): tools/sweep_court.sh <size> > profiles/...
size=${1:-1024}
one() { python3 bench.py --no-cpu --no-exact-leg --repeats 1 --model court --size $size 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s %9.0f Mcs/s %8.3f us/tick' % ('$1', d['value'], d['ms_per_step']*1000))"; }
FIBHIP_NO_MULTI=1 one "one tick per launch"
one "default (shape chosen by measurement)"
FIBHIP_COURT_AGG=0 one "plain kernels (no aggregates)"
for v in 58,14,-2 58,16,-2 58,18,-2 58,20,-2 58,22,-2 58,24,-2 58,25,-2 58,28,-2 58,26,-3 58,12,-1 32,32,256; do FIBHIP_COURT_MULTI3=$v one "K=3 $v"; done
for v in 60,12,-2 60,14,-2 60,16,-2 60,18,-2 60,22,-2 60,30,-2 64,16,256; do FIBHIP_COURT_MULTI3=0,0,0 FIBHIP_COURT_MULTI2=$v one "K=2 $v (no K=3)"; done
