#!/bin/bash
# Shard preview of config 5 with more auxiliary streams (build variants aux7) and more hardware queues (GPU_MAX_HW_QUEUES)
cd "$(dirname "$0")/.."; ROOT=$PWD
for q in 4 8 12; do for v in default aux7; do
  if [ $v = default ]; then unset STRINGZILLAS_ROCM_LIBRARY; else export STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/$v/libstringzillas_rocm_shared.so; fi
  echo "== $v queues $q"; GPU_MAX_HW_QUEUES=$q python scripts/measure_shard_of.py --config 5 --shards 1,4,8 2>&1 | grep "^{" | cut -c1-150
done; done
