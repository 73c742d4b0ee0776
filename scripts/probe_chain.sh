#!/bin/bash
# probe_chain.sh - torch-free GPU visit for hip/myers_chain.hip: parity on awkward candidate counts (a workgroup shares
# one match-mask table between up to 16 candidates), then throughput per wavefronts-per-workgroup setting.
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=${PROBE_ALARM:-40} SZS_ROCM_TIER=chain
run() { echo "--- tier=$SZS_ROCM_TIER waves=${SZS_ROCM_CHAIN_WAVES:-auto} $*"; timeout 90 $P "$@" 2>&1 | tail -${TAIL:-1}; }
for w in 4 8 16; do
    export SZS_ROCM_CHAIN_WAVES=$w
    run lev 1 1 300 500 2
    run lev 5 7 0 40 2
    run lev 3 5 2040 2056 2
    run lev 9 13 1000 5000 2
    run lev 64 63 300 500 2
    run lev 40 37 1800 2300 2
done
unset SZS_ROCM_CHAIN_WAVES
run lev 1 1 90000 110000 2
export PROBE_NO_ORACLE=1
for shape in "16 16 3072 5120" "32 32 4000 4200" "64 64 4000 4200" "128 128 4000 4200" "256 256 4000 4200" "512 512 2500 2600" "64 64 16000 16400"; do
    for w in 4 8 16 auto; do
        if [ $w = auto ]; then unset SZS_ROCM_CHAIN_WAVES; else export SZS_ROCM_CHAIN_WAVES=$w; fi
        run lev $shape 3
    done
done
