#!/bin/bash
# profile_config.sh TAG CONFIG [REPEATS] - kernel stats + PMC passes of one BASELINE.json config at full size
# (scripts/measure_configs.py), same recipe and same corrections as profile_gpu.sh.  Output: gpurun_out/TAG/.
set -u
TAG=$1; CONFIG=$2; REPEATS=${3:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=(python "$ROOT/scripts/measure_configs.py" --configs "$CONFIG" --repeats "$REPEATS")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- "${CMD[@]}" > "$OUT/stats.log" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$pass" | awk '{print $1}')
    rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/pmc_$name" -o pmc -- "${CMD[@]}" > "$OUT/pmc_$name.log" 2>&1 \
        || echo "pmc pass $name failed"
done
python "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"
cat "$OUT/pmc_summary.json" | head -c 4000
grep -h "szs_hip" "$OUT"/stats/*kernel_stats.csv | cut -c1-60,200-330
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete; find "$OUT" -name "*.db" -delete
