#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/${1:-quick}; mkdir -p "$OUT"; cd "$ROOT"
shift || true
SECONDS=0
python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "exit $? in $SECONDS s"
python - "$OUT/bench.json" <<'PY'
import json, sys
line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], "frac", line["roofline"]["frac"], "fresh", (line.get("fresh_batches") or {}).get("value"), "cpu", line.get("cpu_baseline", {}).get("value"))
for record in line.get("configs", []):
    print("   cfg", record.get("config"), record.get("value"), "kernel", record.get("kernel_gcups"), "ms", record.get("ms_per_step"), "cpu", (record.get("cpu_baseline") or {}).get("value"), record.get("error", ""))
PY
