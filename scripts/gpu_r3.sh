#!/bin/bash
# gpu_r3.sh TAG STEP... - round 3's GPU visits; every step under its own `timeout` so a hung kernel cannot hold the box.
#   team-tests | all-tests | team4 | team3 | share4 | bench | configs
set -u
TAG=${1:-r3}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for step in "$@"; do
  case $step in
    team-tests) timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --durations=5 > "$OUT/team_tests.log" 2>&1; echo "exit $?" >> "$OUT/team_tests.log"; tail -40 "$OUT/team_tests.log";;
    all-tests)  timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 --durations=8 > "$OUT/pytest.log" 2>&1; echo "exit $?" >> "$OUT/pytest.log"; tail -30 "$OUT/pytest.log";;
    team4)      timeout 600 python scripts/measure_team.py --config 4 > "$OUT/team_cfg4.jsonl" 2> "$OUT/team_cfg4.err"; cat "$OUT/team_cfg4.jsonl"; tail -3 "$OUT/team_cfg4.err";;
    team3)      timeout 600 python scripts/measure_team.py --config 3 > "$OUT/team_cfg3.jsonl" 2> "$OUT/team_cfg3.err"; cat "$OUT/team_cfg3.jsonl"; tail -3 "$OUT/team_cfg3.err";;
    share4)     timeout 600 python scripts/measure_team.py --config 4 --shards 8 > "$OUT/team_cfg4_share.jsonl" 2> "$OUT/team_cfg4_share.err"; cat "$OUT/team_cfg4_share.jsonl"; tail -3 "$OUT/team_cfg4_share.err";;
    ops)        timeout 300 scripts/bin/team_ops > "$OUT/team_ops.json" 2> "$OUT/team_ops.err"; cat "$OUT/team_ops.json"; tail -3 "$OUT/team_ops.err";;
    sweep)      timeout 900 python scripts/measure_team_sweep.py > "$OUT/team_sweep.jsonl" 2> "$OUT/team_sweep.err"; cat "$OUT/team_sweep.jsonl"; tail -3 "$OUT/team_sweep.err";;
    bench)      timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 3000 "$OUT/bench.json"; tail -3 "$OUT/bench.err";;
    configs)    timeout 600 python scripts/measure_configs.py --configs 2,3,4,5 > "$OUT/configs.jsonl" 2>&1; cat "$OUT/configs.jsonl";;
    *) echo "unknown step $step";;
  esac
done
