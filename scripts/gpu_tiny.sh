#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/tiny; mkdir -p "$OUT"; cd "$ROOT"
P=tests/native/bin/systolic_probe
export PROBE_NO_ORACLE=1 PROBE_ALARM=60
for len in 8 16 32 64; do echo "--- lev 4096 x 4096 len $len"; $P lev 4096 4096 $len $len 5 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  name=$(echo "$pass" | awk '{print $1}')
  rocprofv3 --kernel-trace --output-format csv --pmc $pass -d "$OUT/pmc_$name" -o pmc -- $ROOT/$P lev 4096 4096 8 8 3 > "$OUT/pmc_$name.log" 2>&1
done
python3 "$ROOT/scripts/pmc_summary.py" "$OUT" | head -60
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $ROOT/$P lev 4096 4096 8 8 3 > /dev/null 2>&1; grep -h "szs_hip" "$OUT"/stats/*kernel_stats.csv | cut -c1-60,140-300
