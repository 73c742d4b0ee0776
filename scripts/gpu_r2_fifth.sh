#!/bin/bash
# Fifth GPU pass: whole suite, config 2 with / without plan re-use and as a build variant, planner kernel time, full bench line.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2e}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -25 "$OUT/pytest.log" | cut -c1-300
line() { python - "$1" <<'PY'
import json, sys
try:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], line.get("planner"), "fresh:", line.get("fresh_batches"))
except Exception as problem:
    print(sys.argv[1], "unreadable:", problem)
PY
}
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python bench.py --extra-configs none --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; line "$OUT/$name.json"; }
V=$ROOT/stringzilla_amd/lib_variants
run cfg2_default X=1 -- --steps 300 --warmup 30
run cfg2_no_reuse SZS_ROCM_REUSE=0 -- --steps 300 --warmup 30
run cfg2_td2 STRINGZILLAS_ROCM_LIBRARY=$V/td2/libstringzillas_rocm_shared.so -- --steps 300 --warmup 30
run cfg3 X=1 -- --config 3 --steps 10 --warmup 2
run cfg4 X=1 -- --config 4 --steps 3 --warmup 1
cd /tmp && export TMPDIR=/tmp
SZS_ROCM_REUSE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --extra-configs none --no-cpu-baseline --steps 100 --warmup 10 > "$OUT/trace_bench.json" 2> "$OUT/trace.log"
grep -h "plan_kernel\|myers_short" "$OUT"/trace/*kernel_stats.csv | cut -c1-60,150-400
cd "$ROOT"
timeout 900 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "bench exit $?"; line "$OUT/bench_full.json"
