#!/bin/bash
# First GPU pass of round 2: the whole gpu-marked suite, a traced short bench, then the full bench line.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2a}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -15 "$OUT/pytest.log"
SZS_ROCM_TRACE=1 timeout 300 python bench.py --steps 10 --warmup 3 --extra-configs none --no-cpu-baseline > "$OUT/bench_traced.json" 2> "$OUT/trace.log"
tail -4 "$OUT/trace.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench exit $?"; tail -c 6000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
