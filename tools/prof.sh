#!/bin/bash
# usage (GPU box): tools/prof.sh <tag> [bench args...]   -- kernel trace + PMC passes for bench.py
# env FIBHIP_VARIANT / FIBHIP_K select the kernel.  Outputs under gpurun_out/prof_<tag>/
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu --steps 300 "$@" > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py --no-cpu --steps 100 "$@" > /dev/null 2> $out/pmc_sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_lds -- python3 bench.py --no-cpu --steps 100 "$@" > /dev/null 2> $out/pmc_lds.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu --steps 100 "$@" > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu --steps 100 "$@" > /dev/null 2> $out/pmc_write.err
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
