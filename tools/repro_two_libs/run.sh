#!/bin/bash
# GPU box: does "a library's first kernel AFTER another library's kernels have run" die under rocprofv3 without any fibhip code?
cd "$(dirname "$0")"
export TMPDIR=/tmp
for order in ab ba bba; do
  echo "== order $order, plain"; ./main $order; echo "rc $?"
  echo "== order $order, under rocprofv3 --kernel-trace"; rocprofv3 --kernel-trace -d /tmp/repro_$order -- ./main $order 2>&1 | tail -5; echo "rc ${PIPESTATUS[0]}"
done
spec=$(ls ../../fib_tf_amd/_spec/libfibhip_br_*.so | head -1)
for order in sb bs bbs; do
  echo "== real libraries, order $order, under rocprofv3 --kernel-trace"; rocprofv3 --kernel-trace -d /tmp/repro_real_$order -- ./main_real ../../fib_tf_amd/libfibhip.so $spec $order 2>&1 | grep -v simple_timer | tail -6; echo "rc ${PIPESTATUS[0]}"
done
