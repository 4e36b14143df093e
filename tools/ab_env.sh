#!/bin/bash
# A/B of (environment, library) pairs on ONE box: tools/ab_env.sh "<bench args>" "ENV=... lib.so" "ENV=... lib.so" ...  (3 alternating rounds)
args=$1; shift
for round in 1 2 3; do
  for item in "$@"; do
    lib=${item##* }; envs=${item% *}; [ "$envs" = "$item" ] && envs=""
    env $envs FIBHIP_LIBRARY=$PWD/$lib python3 bench.py --no-cpu --no-exact-leg --repeats 3 $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-60s %9.0f Mcs/s %8.3f us/tick  events: %.3f us/tick (%d launches / %d ticks)' % ('$item', d['value'], d['ms_per_step']*1000, r['us_per_launch']*r['launches_timed']/r['ticks_timed'], r['launches_timed'], r['ticks_timed']))"
  done
done
