#!/bin/bash
# probe_systolic.sh - torch-free GPU visit: the C probe (tests/native) on the systolic tier, every kernel family,
# single band / band chains / ragged batches / one very long pair.  Seconds, not minutes.
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=${PROBE_ALARM:-25}
run() { echo "--- tier=$SZS_ROCM_TIER $*"; timeout 60 $P "$@" 2>&1 | tail -${TAIL:-3}; }
export SZS_ROCM_TIER=systolic
run lev 1 1 300 500 2
run nw 1 1 1 70 2
run levw 3 9 20 60 2
run sw 5 7 0 9 2
run lev 64 64 300 500 2
run nw 64 64 300 500 2
run sw 64 64 300 500 2
run nw 7 5 500 530 2 -11 -2
run sw 7 5 1000 1100 2 2 -1
run lev 16 16 3072 5120 2
run levw 16 16 3072 5120 2
run nw 16 16 3072 5120 2
run sw 8 8 3072 5120 2
run nw 128 128 800 1200 2
run lev 1 1 90000 110000 2
run nw 1 1 90000 110000 2
export SZS_ROCM_TIER=chain
run lev 1 1 300 500 2
run lev 5 7 0 40 2
run lev 64 64 300 500 2
run lev 3 4 2040 2056 2
run lev 16 16 3072 5120 3
run lev 128 128 800 1200 2
run lev 1 1 90000 110000 2
