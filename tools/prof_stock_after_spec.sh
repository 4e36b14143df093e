#!/bin/bash
# GPU box: the round-2 crash scenario under the profiler, once — a kernel of the stock library launched after kernels of the
# specialised Beeler-Reuter build, in one process, with rocprofv3's kernel trace on.
export TMPDIR=/tmp
out=$PWD/gpurun_out/r03/prof_sas
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "stock_library_kernel_after_specialised" > $out/pytest.log 2> $out/pytest.err
rc=$?
echo "rc=$rc" >> $out/pytest.log
tail -3 $out/pytest.log
grep -h "copy_kernel\|strip_mt_kernel\|strip_kernel" $(find $out -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8
exit $rc
