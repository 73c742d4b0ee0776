#!/bin/bash
# probe_lanes_vs_chain.sh - unit-cost Levenshtein: lanes tier (bit-parallel at any length: 2048-row strips beyond 2048
# bytes) vs the band chain vs the planner's own pick, kernel ms per shape.  Torch-free (tests/native).
P=tests/native/bin/systolic_probe
export PROBE_ALARM=60 PROBE_NO_ORACLE=1 SZS_ROCM_SWAP=0
for shape in "1024 1024 1900 2000" "1024 1024 900 1000" "256 256 1900 2000" "128 128 1900 2000" \
             "1024 1024 2100 2200" "512 512 2500 2600" "256 256 4000 4200" "128 128 4000 4200" "64 64 16000 16400"; do
  for t in lanes chain auto; do
    if [ $t = auto ]; then unset SZS_ROCM_TIER; else export SZS_ROCM_TIER=$t; fi
    echo "--- $t $shape"; timeout 120 $P lev $shape 2 2>&1 | tail -1
  done
done
