# same-box A/B of two TREES: BR cheby 512, Fenton exact 512, Fenton fast 512.  The other tree is a checkout of an earlier commit
# with its libraries built in place:  mkdir -p tools/ab/head_tree && git archive <commit> | tar -x -C tools/ab/head_tree &&
# (cd tools/ab/head_tree && python -c "import __graft_entry__ as g; g.build()")   (tools/ab/ is scratch: not tracked)
one() { (cd $1 && timeout -k 10 300 python bench.py $2 --no-cpu --no-exact-leg --no-config-legs --steps 600 --setup 400 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-24s %-16s %9.1f  %.3f us/tick' % ('$1', '$2', d['value'], d['roofline']['us_per_tick']))"); }
for r in 1 2; do
  for a in "--model br" "--exact" ""; do
    one tools/ab/head_tree "$a"
    one . "$a"
  done
done
