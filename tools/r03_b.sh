# two designed experiments on the rocprofv3 SIGSEGV of `bench.py --model br` (first launch from the stock library after the
# specialised library's kernels): (A) stock code object loaded first, (B) HIP's deferred code-object loading off
export TMPDIR=/tmp
out=$PWD/gpurun_out/r03/sigsegv
mkdir -p $out
B="python3 bench.py --model br --no-cpu --no-exact-leg --no-config-legs --repeats 1 --steps 100"
FIBTF_YARDSTICK_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/A -- $B > $out/A.json 2> $out/A.err; echo "A (stock first) rc=$?"
HIP_ENABLE_DEFERRED_LOADING=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out/B -- $B > $out/B.json 2> $out/B.err; echo "B (deferred loading off) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/C -- $B > $out/C.json 2> $out/C.err; echo "C (as is) rc=$?"
grep -c SIGSEGV $out/A.err $out/B.err $out/C.err
