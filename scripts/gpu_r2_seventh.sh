#!/bin/bash
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2g}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -25 "$OUT/pytest.log" | cut -c1-300
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
run cfg3 X=1 -- --config 3 --steps 10 --warmup 2
run cfg5 X=1 -- --config 5 --steps 10 --warmup 2
run cfg5_aux7 STRINGZILLAS_ROCM_LIBRARY=$V/aux7/libstringzillas_rocm_shared.so -- --config 5 --steps 10 --warmup 2
run cfg6 X=1 -- --config 6 --steps 10 --warmup 2
SZS_ROCM_TRACE=1 python bench.py --config 6 --extra-configs none --no-cpu-baseline --steps 2 --warmup 1 2>&1 >/dev/null | grep "szs call" | tail -2
