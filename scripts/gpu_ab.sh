#!/bin/bash
# gpu_ab.sh CONFIG SHARDS VARIANT... - share-of-config timing with the default library and build variants (lib_variants/NAME)
cd "$(dirname "$0")/.."; ROOT=$PWD; CFG=$1; SHARDS=$2; shift 2
for v in default "$@"; do
  if [ $v = default ]; then unset STRINGZILLAS_ROCM_LIBRARY; else export STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/$v/libstringzillas_rocm_shared.so; fi
  echo "== $v"; python scripts/measure_shard_of.py --config $CFG --shards $SHARDS 2>&1 | grep "^{" | cut -c1-150
done
