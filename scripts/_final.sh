OUT=gpurun_out/r4z; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/gpu_tests.txt 2>&1; tail -4 $OUT/gpu_tests.txt
bash scripts/profile_configs.sh r04_pmc2 2 9 3 4 5 6 7 8 > $OUT/pmc.log 2>&1; tail -2 $OUT/pmc.log
cp gpurun_out/r04_pmc2/pmc_configs.json profiles/r04/pmc_configs.json   # bench.py below reads the fresh counters (same library)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
for c in 3 4 5 6; do timeout 600 python scripts/measure_shard_of.py --config $c --shards 1,2,4,8 2>/dev/null; done > $OUT/shard_preview.jsonl; cat $OUT/shard_preview.jsonl
timeout 600 python scripts/measure_queue.py --config 5 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > $OUT/queue_cfg5.jsonl
timeout 600 python scripts/measure_queue.py --config 6 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > $OUT/queue_cfg5u.jsonl; cat $OUT/queue_cfg5u.jsonl | cut -c1-200
