#!/bin/bash
# Second GPU pass of round 2: suite, launch-latency floor, the three planners on config 2, kernel trace, PMC per config.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2b}
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -25 "$OUT/pytest.log"
scripts/bin/launch_latency > "$OUT/launch_latency.json" 2> "$OUT/launch_latency.err"; cat "$OUT/launch_latency.json"
SHORT=(--extra-configs none --no-cpu-baseline --steps 300 --warmup 30)
python bench.py "${SHORT[@]}" > "$OUT/bench_speculated.json" 2> "$OUT/bench_speculated.err"
SZS_ROCM_SPECULATE=0 python bench.py "${SHORT[@]}" > "$OUT/bench_device_planned.json" 2>> "$OUT/bench_speculated.err"
SZS_ROCM_PLANNER=host python bench.py "${SHORT[@]}" > "$OUT/bench_host_planned.json" 2>> "$OUT/bench_speculated.err"
for f in speculated device_planned host_planned; do python - "$OUT/bench_$f.json" <<'PY'
import json, sys
line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("bench_")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"])
PY
done
bash scripts/profile_configs.sh "$(basename "$OUT")/pmc" 2 3 4 5 6 > "$OUT/profile_configs.log" 2>&1
tail -30 "$OUT/profile_configs.log"
