#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/stamps; mkdir -p "$OUT"; cd "$ROOT"
SZS_ROCM_TRACE=1 SZS_ROCM_REUSE=0 STRINGZILLAS_ROCM_LIBRARY=$ROOT/stringzilla_amd/lib_variants/stamps/libstringzillas_rocm_shared.so python bench.py --extra-configs none --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/b.json" 2> "$OUT/trace.log"
grep "planner phases" "$OUT/trace.log" | tail -8; grep "szs call" "$OUT/trace.log" | tail -4
