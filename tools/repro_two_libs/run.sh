#!/bin/bash
# GPU box: does "a library's first kernel AFTER another library's kernels have run" die under rocprofv3 without any fibhip code?
cd "$(dirname "$0")"
export TMPDIR=/tmp
for order in ab ba bba; do
  echo "== order $order, plain"; ./main $order; echo "rc $?"
  echo "== order $order, under rocprofv3 --kernel-trace"; rocprofv3 --kernel-trace -d /tmp/repro_$order -- ./main $order 2>&1 | tail -5; echo "rc ${PIPESTATUS[0]}"
done
