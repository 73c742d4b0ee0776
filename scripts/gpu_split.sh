#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_round2.py -m gpu -q -k "split" 2>&1 | tail -5
for split in 0 2 4; do echo "== split $split"; SZS_ROCM_SPLIT=$split python scripts/measure_shard_of.py --config 5 --shards 1,4,8 2>&1 | grep "^{"; done
