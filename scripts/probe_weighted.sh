#!/bin/bash
# probe_weighted.sh - torch-free GPU visit for the lanes-tier weighted kernels (hip/weighted.hip, hip/weighted_packed.hip):
# parity on ragged batches against the CPU oracle, then BASELINE configs 3 and 4 at full size (timing only), with the
# 16-bit packed kernel and with the 32-bit one (SZS_ROCM_PACKED=0).
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=${PROBE_ALARM:-60} SZS_ROCM_TIER=lanes
run() { echo "--- packed=${SZS_ROCM_PACKED:-auto} $*"; timeout 120 $P "$@" 2>&1 | tail -1; }
run sw 40 300 50 700 1 -4 -1
run sw 33 257 1 100 1 -3 -3
run sw 9 600 0 45 1 -5 -2
run sw 70 70 900 1100 1 -2 -1
run sw 50 50 0 6 1 -1 -1
run nw 40 300 50 700 1 -4 -1
run nw 33 257 1 100 1 -3 -3
run nw 9 600 0 45 1 -5 -2
run nw 50 50 0 6 1 -1 -1
run nw 64 300 10 40 1 -7 -7
run nw 20 20 1300 1400 1 -6 -2
run nw 20 20 1500 1600 1 -6 -2
run levw 40 300 50 700 1
export PROBE_NO_ORACLE=1
for packed in 1 0; do
    export SZS_ROCM_PACKED=$packed
    run nw 1024 1024 384 640 3 -4 -4
    run nw 1024 1024 384 640 3 -4 -1
    run sw 1024 1024 384 640 3 -4 -4
    run sw 512 512 3072 5120 2 -4 -1
done
