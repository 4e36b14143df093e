#!/bin/bash
# A/B of library builds on ONE box (boxes differ by ~1.5 %): tools/ab.sh "<bench args>" libA.so libB.so ...  (3 alternating rounds)
args=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    FIBHIP_LIBRARY=$PWD/$lib FIBHIP_BR_LIBRARY=${BRLIB:+$PWD/${lib/lib_/br_}} python3 bench.py --no-cpu --no-exact-leg --repeats 3 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-28s %9.0f Mcs/s %8.3f us/tick  launch %.3f us' % ('$lib', d['value'], d['ms_per_step']*1000, d['roofline']['us_per_launch']))"
  done
done
