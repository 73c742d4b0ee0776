#!/bin/bash
# The judged sequence, rehearsed: gpu-marked suite, smoke, the default bench line; plus the two-rank same-device line.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2final}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"; tail -4 "$OUT/pytest.log" | cut -c1-200
grep -h "passed\|failed" gpurun_out/reference_suite.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2> "$OUT/bench_driver_like.err"; echo "bench(20/5) exit $?"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?"
python - "$OUT/bench.json" "$OUT/bench_driver_like.json" <<'PY'
import json, sys
for path in sys.argv[1:]:
    line = json.loads(open(path).read().strip().splitlines()[-1])
    print(path.split("/")[-1], "value", line["value"], "ms", line["ms_per_step"], "kernel_ms", line["roofline"]["kernel_ms"], "overhead", line["host_overhead_ms_per_step"], "fresh", (line.get("fresh_batches") or {}).get("value"), "cpu", line.get("cpu_baseline", {}).get("value"))
    for record in line.get("configs", []):
        print("   cfg", record.get("config"), record.get("value"), "kernel", record.get("kernel_gcups"), "ms", record.get("ms_per_step"), "cpu", (record.get("cpu_baseline") or {}).get("value"), record.get("error", ""))
PY
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --same-device > "$OUT/bench_two_ranks_same_device.json" 2> "$OUT/bench_two_ranks.err"; echo "two ranks exit $?"
tail -c 1500 "$OUT/bench_two_ranks_same_device.json"
