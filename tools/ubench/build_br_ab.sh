#!/bin/bash
# tools/ubench/build_br_ab.sh OUT [extra -D flags ...]: builds tools/ubench/br_mt_ab.hip against the tree's kernels.hpp
# (OUT ending in .s: the device ISA instead of a binary).  KHDIR=<dir> takes kernels.hpp from another tree.
root=$(cd "$(dirname "$0")/../.." && pwd)
inc=$(ls "$root"/fib_tf_amd/_spec/br_table_*.inc | head -1)
kh=${KHDIR:-$root/fib_tf_amd/csrc}/kernels.hpp
out=$1; shift
extra=()
case "$out" in *.s) extra=(-S --cuda-device-only);; esac
exec hipcc -w -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DFIB_ONLY_BR "-DFIB_BR_TABLE_INC=\"$inc\"" "-DKH=\"$kh\"" "$@" "${extra[@]}" "$root/tools/ubench/br_mt_ab.hip" -o "$out"
