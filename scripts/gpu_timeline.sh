#!/bin/bash
# gpu_timeline.sh CONFIG SHARDS - kernel timeline of one call of a 1/SHARDS share of a config
cd "$(dirname "$0")/.."; ROOT=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out/timeline; rm -rf gpurun_out/timeline/*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/timeline -o t -- python scripts/measure_shard_of.py --config $1 --shards $2 > gpurun_out/timeline/run.log 2>&1
f=$(find gpurun_out/timeline -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f ${3:-100}
