#!/bin/bash
# Fourth GPU pass: new / changed tests, then a kernel trace of config 2 to see where the call's time goes on the GPU.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${1:-r2d}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_bench_two_ranks.py "tests/test_gpu_parity.py::test_full_size_configs_whole_rows" -m gpu -q > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -30 "$OUT/pytest.log" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --extra-configs none --no-cpu-baseline --steps 100 --warmup 10 > "$OUT/trace_bench.json" 2> "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics
rows = []
for path in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    with open(path) as handle:
        rows += list(csv.DictReader(handle))
rows = [r for r in rows if "szs_hip" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
plan = [r for r in rows if "plan_kernel" in r["Kernel_Name"]]
score = [r for r in rows if "myers_short" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("plan kernel us: median", statistics.median(map(dur, plan)), "scoring kernel us: median", statistics.median(map(dur, score)))
gaps, turn = [], []
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if "plan_kernel" in a["Kernel_Name"] and "myers_short" in b["Kernel_Name"]: gaps.append(gap)
    if "myers_short" in a["Kernel_Name"] and "plan_kernel" in b["Kernel_Name"]: turn.append(gap)
print("gap plan->score us: median", statistics.median(gaps), " gap score->next plan us (host turnaround): median", statistics.median(turn))
PY
