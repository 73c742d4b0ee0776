#!/bin/bash
# probe_weighted_variants.sh - A/B of build variants of hip/weighted.hip through the C probe (results: profiles/r01/
# weighted_occupancy_variants_v1.txt).  Build the variants first, in-tree, next to the real library:
#   make -C stringzilla_amd/csrc OUT=../lib_variants/w3  EXTRA="-DSZS_WEIGHTED_WAVES=3"
#   make -C stringzilla_amd/csrc OUT=../lib_variants/w3r EXTRA="-DSZS_WEIGHTED_WAVES=3 -DSZS_WEIGHTED_RECOMPUTE=1"
#   make -C stringzilla_amd/csrc OUT=../lib_variants/w2r EXTRA="-DSZS_WEIGHTED_RECOMPUTE=1"
# (stringzilla_amd/lib_variants/ is git-ignored; the probe picks a variant up through LD_LIBRARY_PATH.)
P=tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=60 SZS_ROCM_TIER=lanes
for v in w3 w3r w2r; do
  export LD_LIBRARY_PATH=$PWD/stringzilla_amd/lib_variants/$v
  unset PROBE_NO_ORACLE
  echo "=== $v"; 
  timeout 100 $P sw 40 300 50 700 1 -4 -1 2>&1 | tail -1
  timeout 100 $P nw 40 300 50 700 1 -4 -1 2>&1 | tail -1
  export PROBE_NO_ORACLE=1
  timeout 100 $P sw 512 512 3072 5120 2 -4 -1 2>&1 | tail -1
  timeout 100 $P nw 512 512 3072 5120 2 -4 -1 2>&1 | tail -1
done
