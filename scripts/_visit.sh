OUT=gpurun_out/r4n; mkdir -p $OUT
for c in 3 4 5 6; do timeout 600 python scripts/measure_shard_of.py --config $c --shards 1,2,4,8 2>/dev/null; done > $OUT/shard_preview.jsonl
cat $OUT/shard_preview.jsonl
timeout 600 python scripts/measure_queue.py --config 5 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > $OUT/queue_cfg5.jsonl; cat $OUT/queue_cfg5.jsonl
timeout 600 python scripts/measure_queue.py --config 6 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > $OUT/queue_cfg5u.jsonl; cat $OUT/queue_cfg5u.jsonl
SZS_ROCM_TRACE=1 SZS_ROCM_QUEUE_PRIORITY=0 timeout 300 python scripts/measure_queue.py --config 5 --shards 8 --words auto --seconds 0.004 > $OUT/trace_before.txt 2>&1
timeout 600 python scripts/measure_queue_shapes.py > $OUT/queue_shapes.jsonl 2>/dev/null; tail -3 $OUT/queue_shapes.jsonl
timeout 900 python -m pytest tests/test_bench_two_ranks.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | tail -3
