one() { FIBHIP_BR_LIBRARY=$PWD/fib_tf_amd/_spec/libfibhip_br_$1.so timeout -k 10 300 python bench.py --model br --no-cpu --no-exact-leg --no-config-legs --steps 600 --setup 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 $2 %9.1f  %.3f us/tick' % (d['value'], d['roofline']['us_per_tick']))"; }
for r in 1 2 3; do
 (cd tools/ab/head_tree && timeout -k 10 300 python bench.py --model br --no-cpu --no-exact-leg --no-config-legs --steps 600 --setup 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('head                  %9.1f  %.3f us/tick' % (d['value'], d['roofline']['us_per_tick']))")
 one 36f0e23d7ce74d6f base
 one 4a1e75da35f09e24 lean_poll
 (cd tools/ubench && ./mt_ab_head 32 30 && ./mt_ab_keep 32 30 && ./mt_ab_new 32 30)
done
