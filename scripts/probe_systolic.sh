#!/bin/bash
# probe_systolic.sh - torch-free GPU visit: the C probe (tests/native) on the shapes that separate hypotheses, against
# the product library and, when present, its debugging variants (stringzilla_amd/lib_variants/*).
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0
run() { echo "--- [$LIBTAG] tier=$SZS_ROCM_TIER $*"; timeout 45 $P "$@" 2>&1 | tail -4; echo "rc=$?"; }
LIBTAG=product
export SZS_ROCM_TIER=systolic
run lev 1 1 300 500 2
run nw 1 1 300 500 2
run lev 64 64 300 500 2
run nw 64 64 300 500 2
run levw 3 9 20 60 2
run sw 64 64 300 500 2
run lev 16 16 3072 5120 3
run nw 16 16 3072 5120 2
run sw 16 16 3072 5120 2
run levw 16 16 3072 5120 2
for variant in stringzilla_amd/lib_variants/*/; do
    LIBTAG=$(basename $variant)
    export LD_LIBRARY_PATH=$PWD/$variant
    run lev 16 16 3072 5120 3
    run lev 64 64 300 500 1
done
unset LD_LIBRARY_PATH
LIBTAG=product
export SZS_ROCM_TIER=lanes
run lev 64 64 300 500 1
