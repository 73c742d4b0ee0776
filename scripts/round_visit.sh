#!/bin/bash
# round_visit.sh ROUND [TAG] - ONE GPU visit that produces what profiles/rROUND cites (about fifteen minutes of box time):
#   the whole GPU suite, kernel stats + PMC passes of every bench config (scripts/profile_configs.sh), the default bench line and
#   the driver's 20-step line (read against the counters just taken: same library, `pmc_stale` false), smoke(), the shard previews
#   and the one-launch kernel against the per-width launches at every share, the tiny-token launch from plain C (parity + words of text) and
#   the reference's tokeniser on the prose in this image.  Everything lands under gpurun_out/TAG; the PMC
#   summary is also copied to profiles/rROUND on the box so that bench.py finds it - copy the files you cite into profiles/.
ROUND=${1:-04}; TAG=${2:-r${ROUND}_final}
OUT=gpurun_out/$TAG; mkdir -p "$OUT" "profiles/r$ROUND"
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > "$OUT/gpu_tests.txt" 2>&1; tail -4 "$OUT/gpu_tests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.txt" 2>&1; tail -2 "$OUT/smoke.txt"
bash scripts/profile_configs.sh "${TAG}_pmc" 2 9 3 4 5 6 7 8 10 12 11 > "$OUT/pmc.log" 2>&1; tail -2 "$OUT/pmc.log"
cp "gpurun_out/${TAG}_pmc/pmc_configs.json" "profiles/r$ROUND/pmc_configs.json"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 300 "$OUT/bench_default.json"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_20.json" 2> "$OUT/bench_20.err"
for c in 3 4 5 6; do timeout 600 python scripts/measure_shard_of.py --config $c --shards 1,2,4,8 2>/dev/null; done > "$OUT/shard_preview.jsonl"; cat "$OUT/shard_preview.jsonl"
timeout 600 python scripts/measure_queue.py --config 5 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > "$OUT/queue_cfg5.jsonl"
timeout 600 python scripts/measure_queue.py --config 6 --shards 1,2,4,8 --words auto --seconds 0.4 2>/dev/null > "$OUT/queue_cfg5u.jsonl"; cut -c1-200 "$OUT/queue_cfg5u.jsonl"
bash scripts/measure_words.sh "$TAG/words" > /dev/null 2>&1; grep -c '"failures": 0' "$OUT/words/words.txt"; grep "tiny kernel" "$OUT/words/words.txt" | tail -2
bash scripts/run_real_text.sh > "$OUT/real_text.log" 2>&1; cp gpurun_out/real_text/real_text.jsonl "$OUT/real_text.jsonl"; cut -c1-220 "$OUT/real_text.jsonl"
# round 6: the C host under ASan + UBSan driven by the torch-free probes (scripts/gpu_visit.sh asan), and the harness race of
# scripts/repro_plain_path.py (a matrix filled on torch's stream, scored on the scope's) with and without the fill drained
bash scripts/gpu_visit.sh "$TAG" asan > "$OUT/asan_visit.log" 2>&1; tail -3 "$OUT/asan_visit.log"
for mode in sync nosync; do timeout 600 python scripts/repro_plain_path.py 40 $mode 2>&1 | tail -3 | cut -c1-300; done > "$OUT/harness_stream_race.txt"; tail -2 "$OUT/harness_stream_race.txt"
