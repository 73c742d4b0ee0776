#!/bin/bash
# Configs 5u and 5 with more auxiliary streams (build variants aux8, aux11) and more hardware queues (GPU_MAX_HW_QUEUES)
cd "$(dirname "$0")/.."; ROOT=$PWD
for q in 8 12; do for v in default aux8 aux11; do
  if [ $v = default ]; then unset STRINGZILLAS_ROCM_LIBRARY; else export STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/$v/libstringzillas_rocm_shared.so; fi
  for c in 6 5; do echo "== $v queues $q cfg $c"; GPU_MAX_HW_QUEUES=$q python scripts/measure_shard_of.py --config $c --shards 1,8 2>&1 | grep "^{" | cut -c1-120; done
done; done
