#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r2i; mkdir -p "$OUT"; cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -12 "$OUT/pytest.log" | cut -c1-300
SZS_ROCM_TRACE=1 python bench.py --config 6 --extra-configs none --no-cpu-baseline --steps 3 --warmup 1 2>&1 >/dev/null | grep "szs call" | tail -2
python bench.py --config 6 --extra-configs none --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg6', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['host_overhead_ms_per_step'])"
