#!/bin/bash
# probe_myers_long_occupancy.sh - the long Myers kernels at one vs two wavefronts per SIMD (results: profiles/r01/
# myers_long_occupancy_v1.txt; two is the default since).  `base` is the in-tree library; build the other side with
#   make -C stringzilla_amd/csrc OUT=../lib_variants/long2 EXTRA="-DSZS_MYERS_LONG_WAVES=2"      (or =1, to compare back)
P=tests/native/bin/systolic_probe
export PROBE_ALARM=60 PROBE_NO_ORACLE=1 SZS_ROCM_SWAP=0 SZS_ROCM_TIER=lanes
for v in base long2; do
  if [ $v = base ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$PWD/stringzilla_amd/lib_variants/$v; fi
  for shape in "1024 1024 1900 2000" "1024 1024 1600 1700" "1024 1024 1300 1500" "3000 300 8 2048"; do
    echo "--- $v $shape"; timeout 120 $P lev $shape 2 2>&1 | tail -1
  done
done
unset PROBE_NO_ORACLE; export LD_LIBRARY_PATH=$PWD/stringzilla_amd/lib_variants/long2
echo "--- parity long2"; timeout 120 $P lev 12 300 1500 2048 1 2>&1 | tail -1
