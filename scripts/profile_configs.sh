#!/bin/bash
# profile_configs.sh TAG [CONFIG ...] - per BASELINE.json config, the SAME command bench.py's `configs` records time
# (`bench.py --config N --extra-configs none`), under
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/TAG/cfgN/stats/*kernel_stats.csv   (average durations)
#   2. PMC passes, each in its own run, never mixed with trace domains other than --kernel-trace
#                                                  -> gpurun_out/TAG/cfgN/pmc_*/
# and folds everything with scripts/pmc_configs.py -> gpurun_out/TAG/pmc_configs.json (copy it to profiles/rNN/).
set -u
TAG=${1:-pmc_configs}; shift || true
CONFIGS=("$@"); [ ${#CONFIGS[@]} -eq 0 ] && CONFIGS=(3 4 5 6)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for N in "${CONFIGS[@]}"; do
    case $N in 2) STEPS=(--steps 50 --warmup 5);; 4) STEPS=(--steps 2 --warmup 1);; *) STEPS=(--steps 5 --warmup 1);; esac
    ARGS=(--config "$N" --extra-configs none --no-cpu-baseline "${STEPS[@]}")
    [ "$N" = 11 ] && ARGS=(--fingerprints-only --extra-seconds 0.05) # `szs_fingerprints_u32tape` alone: cfg11:fingerprint_segments_kernel
    D=$OUT/cfg$N; mkdir -p "$D"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o stats -- python "$ROOT/bench.py" "${ARGS[@]}" > "$D/bench.json" 2> "$D/stats.log" \
        || echo "cfg$N stats run failed"
    tail -c 400 "$D/bench.json"; echo
    for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
                "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
                "FETCH_SIZE" "WRITE_SIZE"; do
        name=$(echo "$pass" | awk '{print $1}')
        timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$D/pmc_$name" -o pmc -- python "$ROOT/bench.py" "${ARGS[@]}" > "$D/pmc_$name.log" 2>&1 \
            || echo "cfg$N pmc pass $name failed (see $D/pmc_$name.log)"
    done
    find "$D" -name "*_kernel_trace.csv" -size +2M -delete
    find "$D" -name "*.db" -delete
done
python3 "$ROOT/scripts/pmc_configs.py" "$OUT" > "$OUT/pmc_configs.json" 2> "$OUT/pmc_configs.err"
head -c 3000 "$OUT/pmc_configs.json"; tail -3 "$OUT/pmc_configs.err"
du -sh "$OUT"
