#!/bin/bash
# probe_fingerprints.sh - torch-free GPU visit for `szs_fingerprints_*`: parity against the CPU oracle, then throughput.
P=tests/native/bin/fingerprints_probe
run() { echo "--- $*"; timeout 120 $P "$@" 2>&1 | tail -${TAIL:-4}; }
run 512 20 0 300 2
run 64 50 0 40 1 3
run 100 30 0 200 1
run 7 10 0 100 1 3 5
run 1024 8 3000 13000 2
run 128 6 4090 4100 1 4 7
run 192 5 0 9000 1 2 5 33
PROBE_BINARY=1 run 256 12 0 5000 1 3 1000
export PROBE_NO_ORACLE=1
run 1024 20000 100 100 3
run 1024 2000 1000 1000 3
run 1024 200 10000 10000 3
run 1024 20 100000 100000 3
