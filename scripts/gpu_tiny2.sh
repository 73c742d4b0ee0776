#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
export PROBE_NO_ORACLE=1 PROBE_ALARM=60
for v in default mid; do
  if [ $v = default ]; then export LD_LIBRARY_PATH=$ROOT/stringzilla_amd/lib; else export LD_LIBRARY_PATH=$ROOT/stringzilla_amd/lib_variants/$v; fi
  for shape in "1 14" "4 7" "8 8" "1 40" "20 60"; do echo "--- $v lev 4096 x 4096 len $shape: $(tests/native/bin/systolic_probe lev 4096 4096 $shape 4 2>&1 | tail -1)"; done
done
unset LD_LIBRARY_PATH
bash scripts/gpu_variant_cfg.sh 2 mid
