#!/bin/bash
# probe_systolic.sh - torch-free GPU visit: the C probe (tests/native) on the shapes that separate hypotheses, against
# the product library and, when present, its debugging variants (stringzilla_amd/lib_variants/*).
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=8
run() { echo "--- [$LIBTAG] tier=$SZS_ROCM_TIER $*"; timeout 20 $P "$@" 2>&1 | tail -${TAIL:-4}; }
LIBTAG=product
export SZS_ROCM_TIER=systolic
run lev 1 1 300 500 2
run nw 1 1 300 500 2
run levw 3 9 20 60 2
run lev 64 64 300 500 2
run lev 16 16 3072 5120 3
run levw 16 16 3072 5120 2
run nw 16 16 3072 5120 2
run sw 16 16 3072 5120 2
run lev 1 1 90000 110000 2
if [ -d stringzilla_amd/lib_variants/trace ]; then
    LIBTAG=trace
    export LD_LIBRARY_PATH=$PWD/stringzilla_amd/lib_variants/trace
    TAIL=12 run lev 1 1 300 500 1
    TAIL=40 run lev 16 16 3072 5120 1
    unset LD_LIBRARY_PATH
fi
