#!/bin/bash
# run_check.sh TAG [full] - the round's standard GPU visit: parity tests, per-shape timings, the bench line (and, with
# `full`, the unified-memory and per-config timings).  Every step runs under its own `timeout` so that a hung kernel
# cannot hold the box until gpurun's limit.
set -u
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest exit $?" >> "$OUT/pytest.log"
tail -30 "$OUT/pytest.log"
timeout 300 python scripts/measure_shapes.py > "$OUT/shapes.jsonl" 2> "$OUT/shapes.err"; echo "shapes exit $?"; cat "$OUT/shapes.jsonl"; tail -3 "$OUT/shapes.err"
timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 2500 "$OUT/bench.json"
if [ "${2:-}" = full ]; then
    timeout 200 python scripts/measure_unified.py > "$OUT/unified.jsonl" 2> "$OUT/unified.err"; echo "unified exit $?"; cat "$OUT/unified.jsonl"; tail -3 "$OUT/unified.err"
    timeout 300 python scripts/measure_configs.py --configs 2,3,4,5 > "$OUT/configs.jsonl" 2>&1; cat "$OUT/configs.jsonl"
fi
