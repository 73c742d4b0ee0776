#!/bin/bash
# pmc_weighted.sh TAG - PMC passes (own runs, --kernel-trace only) over ONE full-size config-4 call of the weighted lanes
# kernel through the torch-free C probe; scripts/pmc_summary.py folds them  -> gpurun_out/TAG/pmc_summary.json
set -u
TAG=${1:-pmc_weighted}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
P=$ROOT/tests/native/bin/systolic_probe
export SZS_ROCM_SWAP=0 PROBE_ALARM=120 SZS_ROCM_TIER=lanes PROBE_NO_ORACLE=1
cd /tmp && export TMPDIR=/tmp
FAMILY=${FAMILY:-sw}
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    name=$(echo "$pass" | awk '{print $1}')
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/pmc_$name" -o pmc -- $P $FAMILY 512 512 3072 5120 1 -4 -1 > "$OUT/pmc_$name.log" 2>&1 \
        || echo "pmc pass $name failed (see $OUT/pmc_$name.log)"
    tail -1 "$OUT/pmc_$name.log"
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"
cat "$OUT/pmc_summary.json" | head -c 4000
find "$OUT" -name "*.db" -delete
