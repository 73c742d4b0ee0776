#!/bin/bash
# gpu_visit.sh TAG STEP... - the GPU visits of a round; every step under its own `timeout` so a hung kernel cannot hold the box.
#   team-tests | all-tests | team4 | team3 | share4 | sweep | ops | asan | bench | configs | timeline5 | timeline6 | shapes | realtext | previews
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
    asan)       # the C host under ASan + UBSan, driven by the torch-free probes: every family, the chained tiers, the node driver
                # (use_sigaltstack=0: the HIP runtime pins pages the sanitizer wants to unmap when a worker thread exits)
                export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0 UBSAN_OPTIONS=print_stacktrace=1 PROBE_ALARM=120
                # (the sanitized library is built HERE, not by build(): `make -C stringzilla_amd/csrc asan && make -C tests/native asan` - say so when it is older than the sources)
                [ -n "$(find stringzilla_amd/csrc/host stringzilla_amd/csrc/hip -newer stringzilla_amd/lib_asan/libstringzillas_rocm_shared.so -type f 2>/dev/null | head -1)" ] && echo "WARNING: stringzilla_amd/lib_asan is older than the sources it was built from"
                { for args in "lev 40 300 50 700 1" "levw 40 300 50 700 1" "nw 40 300 50 700 1" "sw 40 300 50 700 1" "nw 9 300 900 1100 1 -4 -1" "lev 3 4 2040 2100 1" "lev 1 1 30000 40000 1" "nw 1 1 30000 40000 1"; do
                    echo "--- systolic_probe $args"; timeout 300 tests/native/bin/systolic_probe_asan $args 2>&1 | grep -v "^    #" | tail -6; done
                  for args in "lev 300 700 40 200 6" "lev 1024 1024 96 160 4"; do # round 5: streams of fresh batches - the launch that plans itself
                    echo "--- systolic_probe $args (PROBE_ALTERNATE=1)"; PROBE_ALTERNATE=1 timeout 300 tests/native/bin/systolic_probe_asan $args 2>&1 | grep -v "^    #" | tail -7; done
                  for args in "lev 3000 2500 10 120 4" "lev 1500 5000 1 60 3" "lev 6000 3000 12 40 3"; do # round 6: sides beyond 1024 strings plan themselves too (two walks over the offsets; merged candidate blocks)
                    echo "--- systolic_probe $args (PROBE_ALTERNATE=1)"; PROBE_ALTERNATE=1 timeout 300 tests/native/bin/systolic_probe_asan $args 2>&1 | grep -v "^    #" | tail -7; done
                  for args in "lev 600 2100 1 12 4" "lev 300 700 0 20 4"; do # ... and tiny tokens straight from the tapes (the second with a fifth of its strings beyond 16 bytes: the longer tokens of the same launch)
                    echo "--- systolic_probe $args (PROBE_ALTERNATE=1 SZS_ROCM_TINY=1)"; PROBE_ALTERNATE=1 SZS_ROCM_TINY=1 timeout 300 tests/native/bin/systolic_probe_asan $args 2>&1 | grep -v "^    #" | tail -5; done
                  for args in "mix:60:40 100 700 3" "mix:100:255 300 1000 3 7" "mix:1000:200 40 300 2"; do # ... the one launch of tiny and longer tokens (refusals included: tiny = 1)
                    for knob in 1 2; do echo "--- words_probe $args (SZS_ROCM_TINY=$knob)"; SZS_ROCM_TINY=$knob timeout 300 tests/native/bin/words_probe_asan $args 2>&1 | grep -v "^    #" | tail -4; done; done
                  for mode in "PROBE_UTF8=1" "PROBE_UTF8=1 PROBE_WIDE=1" "PROBE_SYMMETRIC=1" "PROBE_UTF8=1 PROBE_SYMMETRIC=1"; do # round 6: the codepoint twin (one pass writes the runes as bytes), symmetric calls of words
                    for args in "mix:40:60 300 1100 4 5" "mix:150:255 1100 1100 3"; do echo "--- words_probe $args ($mode SZS_ROCM_TINY=1)"; env $mode SZS_ROCM_TINY=1 timeout 300 tests/native/bin/words_probe_asan $args 2>&1 | grep -v "^    #" | tail -5; done; done
                  for args in "lev 70 300 10 400 0 0" "nw 33 200 100 600 0 0 0" "sw 20 100 500 900 0" "lev-sym 150 0 10 400 0 0 0" "nw-sym 90 0 100 600 0 0 0 0 0 0 0 0" "lev 5 40 10 90 0 0 0 0 0 0 0 0"; do
                    echo "--- node_probe $args"; timeout 300 tests/native/bin/node_probe_asan $args 2>&1 | grep -v "^    #" | tail -6; done
                  for args in "lev 120 900 8 2040 0" "lev-sym 200 0 8 2040 0 0"; do # the one-launch kernel: its queue is planned by the sanitized host
                    echo "--- node_probe $args (SZS_ROCM_QUEUE=1)"; SZS_ROCM_QUEUE=1 timeout 300 tests/native/bin/node_probe_asan $args 2>&1 | grep -v "^    #" | tail -6; done
                  echo "--- fingerprints_probe"; timeout 300 tests/native/bin/fingerprints_probe_asan 64 50 100 5000 1 2>&1 | tail -4; } > "$OUT/asan.log" 2>&1
                grep -c "ERROR: AddressSanitizer\|runtime error" "$OUT/asan.log"; tail -40 "$OUT/asan.log";;
    bench)      timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 3000 "$OUT/bench.json"; tail -3 "$OUT/bench.err";;
    configs)    timeout 600 python scripts/measure_configs.py --configs 2,3,4,5 > "$OUT/configs.jsonl" 2>&1; cat "$OUT/configs.jsonl";;
    timeline5)  timeout 300 bash scripts/gpu_timeline.sh 5 8 > "$OUT/timeline_cfg5_eighth.txt" 2>&1; tail -30 "$OUT/timeline_cfg5_eighth.txt";;
    timeline6)  timeout 300 bash scripts/gpu_timeline.sh 6 8 > "$OUT/timeline_cfg6_eighth.txt" 2>&1; tail -30 "$OUT/timeline_cfg6_eighth.txt";;
    shapes)     timeout 900 python scripts/measure_shapes.py > "$OUT/shapes.jsonl" 2> "$OUT/shapes.err"; cat "$OUT/shapes.jsonl"; tail -3 "$OUT/shapes.err";;
    realtext)   rm -f gpurun_out/real_text/real_text.jsonl; timeout 600 bash scripts/run_real_text.sh > "$OUT/real_text.log" 2>&1; cp gpurun_out/real_text/real_text.jsonl "$OUT/real_text.jsonl"; cat "$OUT/real_text.jsonl";;
    previews)   for config in 3 4 5 6; do timeout 300 python scripts/measure_shard_of.py --config $config --shards 1,2,4,8; done > "$OUT/shard_preview.jsonl" 2> "$OUT/shard_preview.err"; cat "$OUT/shard_preview.jsonl";;
    bench20)    timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_20.json" 2> "$OUT/bench_20.err"; tail -c 1500 "$OUT/bench_20.json"; tail -3 "$OUT/bench_20.err";;
    pmc)        timeout 1500 bash scripts/profile_configs.sh $TAG/pmc 2 3 4 5 6 7 8 > "$OUT/pmc.log" 2>&1; tail -5 "$OUT/pmc.log";;
    queue-tests) timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -q -x --durations=5 > "$OUT/queue_tests.log" 2>&1; echo "exit $?" >> "$OUT/queue_tests.log"; tail -40 "$OUT/queue_tests.log";;
    lev-tests)  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -k "levenshtein or config or known or golden or cross_product or symmetric or closed or input_formats or callback" --durations=5 > "$OUT/lev_tests.log" 2>&1; echo "exit $?" >> "$OUT/lev_tests.log"; tail -30 "$OUT/lev_tests.log";;
    queue5)     timeout 600 python scripts/measure_queue.py --config 5 --shards 1,8 > "$OUT/queue_cfg5.jsonl" 2> "$OUT/queue_cfg5.err"; cat "$OUT/queue_cfg5.jsonl"; tail -3 "$OUT/queue_cfg5.err";;
    queue-shapes) timeout 600 python scripts/measure_queue_shapes.py > "$OUT/queue_shapes.jsonl" 2> "$OUT/queue_shapes.err"; cat "$OUT/queue_shapes.jsonl"; tail -3 "$OUT/queue_shapes.err";;
    *) echo "unknown step $step";;
  esac
done
