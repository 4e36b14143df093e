#!/bin/bash
# usage (GPU box): tools/prof.sh <tag> [bench args...]   -- kernel trace + PMC passes for bench.py
# env FIBHIP_VARIANT / FIBHIP_K select the kernel.  Outputs under gpurun_out/prof_<tag>/ ; the condensed record
# (summary.txt, counters.json) is what gets copied into profiles/.
# Counter passes run on their own (no trace domains next to --pmc), each within the per-block slot budget of
# MI355X_MICROARCH.md "rocprofv3 PMC slots" (SQ 8, TCC 4 with FETCH_SIZE = 3 and WRITE_SIZE = 2, GRBM 2).
set -u
# one plan for every pass: a counter pass slows the kernels down and the first tick's measurement then picks another tile
# shape than the trace pass did (seen in round 3: 44x27 under --pmc, 44x25 under the trace).  The rule-based plan is what
# the measurement picks at 512x512 / 1024x1024; pass FIBHIP_VARIANT for any other shape (Fenton 4096x4096: 5,54,28,-3).
if [ -z "${FIBHIP_VARIANT:-}" ]; then export FIBHIP_AUTOTUNE=${FIBHIP_AUTOTUNE:-0}; fi
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --no-cpu --no-exact-leg --no-config-legs --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B --steps 300 "$@" > $out/bench_trace.json 2> $out/trace.err
pass() {   # name counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- $B --steps 100 --setup 50 "${ARGS[@]}" > $out/$name.json 2> $out/$name.err || echo "pass $name failed (see $name.err)"
}
ARGS=("$@")
pass pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass pmc_trans SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass pmc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
pass pmc_eaw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum
python3 tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
