#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2h}
mkdir -p "$OUT"
cd "$ROOT"
line() { python - "$1" <<'PY'
import json, sys
try:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], line.get("planner"))
except Exception as problem:
    print(sys.argv[1], "unreadable:", problem)
PY
}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python bench.py --extra-configs none --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; line "$OUT/$name.json"; }
V=$ROOT/stringzilla_amd/lib_variants
run cfg4 X=1 -- --config 4 --steps 3 --warmup 1
run cfg4_aff2 STRINGZILLAS_ROCM_LIBRARY=$V/aff2/libstringzillas_rocm_shared.so -- --config 4 --steps 3 --warmup 1
scripts/bin/valu_peak > "$OUT/valu_peak.json" 2> "$OUT/valu_peak.err" || echo "valu_peak failed"
scripts/bin/launch_latency > "$OUT/launch_latency.json" 2>/dev/null; cat "$OUT/launch_latency.json"
bash scripts/profile_configs.sh "$(basename "$OUT")/pmc" 2 3 4 5 6 > "$OUT/profile_configs.log" 2>&1
tail -5 "$OUT/profile_configs.log" | cut -c1-300
