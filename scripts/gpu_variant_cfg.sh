#!/bin/bash
# gpu_variant_cfg.sh CONFIG VARIANT... - bench one config with the default library and with build variants (lib_variants/NAME)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; CFG=$1; shift
STEPS=(--steps 10 --warmup 2); [ "$CFG" = 4 ] && STEPS=(--steps 3 --warmup 1); [ "$CFG" = 2 ] && STEPS=(--steps 300 --warmup 30)
for v in default "$@"; do
  if [ $v = default ]; then unset STRINGZILLAS_ROCM_LIBRARY; else export STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/$v/libstringzillas_rocm_shared.so; fi
  python bench.py --config $CFG --extra-configs none --no-cpu-baseline "${STEPS[@]}" | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$CFG $v', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['results_checksum'])"
done
