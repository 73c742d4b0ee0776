P=tests/native/bin/systolic_probe
export PROBE_ALARM=60 PROBE_NO_ORACLE=1 SZS_ROCM_SWAP=0
for shape in "1024 1024 1900 2000" "1024 1024 900 1000" "1024 1024 400 500" "2048 2048 250 300" "256 256 1900 2000" "128 128 1900 2000"; do
  for t in lanes chain auto; do
    if [ $t = auto ]; then unset SZS_ROCM_TIER; else export SZS_ROCM_TIER=$t; fi
    echo "--- $t $shape"; timeout 120 $P lev $shape 2 2>&1 | tail -1
  done
done
