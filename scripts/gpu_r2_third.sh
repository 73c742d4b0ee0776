#!/bin/bash
# Third GPU pass: suite (incl. the reference's own), planner fix, packed profile layout, stream fan-out, build variants.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2c}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -30 "$OUT/pytest.log"; tail -40 gpurun_out/reference_suite.log 2>/dev/null | cut -c1-300
cp gpurun_out/reference_suite.log "$OUT/" 2>/dev/null
line() { python - "$1" <<'PY'
import json, sys
try:
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], "checksum", line["results_checksum"])
except Exception as problem:
    print(sys.argv[1], "unreadable:", problem)
PY
}
run() { # run NAME [ENV=VALUE ...] -- bench args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python bench.py --extra-configs none --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    line "$OUT/$name.json"
}
V=$ROOT/stringzilla_amd/lib_variants
run cfg2_default X=1 -- --steps 300 --warmup 30
run cfg2_mid STRINGZILLAS_ROCM_LIBRARY=$V/mid/libstringzillas_rocm_shared.so -- --steps 300 --warmup 30
run cfg2_td2 STRINGZILLAS_ROCM_LIBRARY=$V/td2/libstringzillas_rocm_shared.so -- --steps 300 --warmup 30
run cfg3_default X=1 -- --config 3 --steps 10 --warmup 2
run cfg4_default X=1 -- --config 4 --steps 3 --warmup 1
run cfg4_aff4 STRINGZILLAS_ROCM_LIBRARY=$V/aff4/libstringzillas_rocm_shared.so -- --config 4 --steps 3 --warmup 1
run cfg5_streams X=1 -- --config 5 --steps 10 --warmup 2
run cfg5_one_stream SZS_ROCM_STREAMS=0 -- --config 5 --steps 10 --warmup 2
run cfg6_streams X=1 -- --config 6 --steps 10 --warmup 2
run cfg6_one_stream SZS_ROCM_STREAMS=0 -- --config 6 --steps 10 --warmup 2
