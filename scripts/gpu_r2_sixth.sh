#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2f}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -q -x > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -15 "$OUT/pytest.log" | cut -c1-300
line() { python - "$1" <<'PY'
import json, sys
try:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], line.get("planner"), "fresh:", (line.get("fresh_batches") or {}).get("value"), (line.get("fresh_batches") or {}).get("ms_per_step"))
except Exception as problem:
    print(sys.argv[1], "unreadable:", problem)
PY
}
python bench.py --extra-configs none --no-cpu-baseline --steps 300 --warmup 30 > "$OUT/cfg2.json" 2> "$OUT/cfg2.err"; line "$OUT/cfg2.json"
cd /tmp && export TMPDIR=/tmp
SZS_ROCM_REUSE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --extra-configs none --no-cpu-baseline --steps 100 --warmup 10 > "$OUT/trace_bench.json" 2> "$OUT/trace.log"
grep -h "plan_kernel\|myers_short" "$OUT"/trace/*kernel_stats.csv | cut -c1-50,150-400
