#!/bin/bash
# which launch shape the first tick's measurement picks, per model and grid size (GPU box): tools/chosen_plans.sh > profiles/...
export FIBHIP_PRINT_PLAN=1
for m in fenton br court; do
  for n in 384 512 640 768 1024 1536 2048 4096; do
    [ $m != fenton ] && [ $n -gt 2048 ] && continue
    python3 bench.py --no-cpu --no-exact-leg --repeats 1 --steps 50 --setup 20 --model $m --size $n 2>&1 >/dev/null | grep "^fibhip:" | sort -u
  done
done
