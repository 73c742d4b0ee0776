#!/usr/bin/env python3
"""pmc_summary.py DIR - folds rocprofv3 counter-collection CSVs (one sub-directory per --pmc pass, written by
scripts/profile_gpu.sh) into per-kernel, per-dispatch averages, and applies the gfx950 unit corrections of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section):

    FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KILOBYTES;
    on gfx950 FETCH_SIZE tallies 128-byte requests of wide coalesced streaming reads at 64 bytes (x2 for those);
    other access widths and WRITE_SIZE are uncalibrated - both the raw and the doubled fetch figure are printed.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
per_kernel = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as handle:
        for row in csv.DictReader(handle):
            name = row.get("Kernel_Name", "")
            if "szs_hip" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("szs_hip::", "")
            per_kernel[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            per_kernel[short]["_vgpr"] = [float(row.get("VGPR_Count", 0) or 0)]
            per_kernel[short]["_sgpr"] = [float(row.get("SGPR_Count", 0) or 0)]
            per_kernel[short]["_grid"].append(float(row.get("Grid_Size", 0) or 0))

summary = {}
for kernel, counters in sorted(per_kernel.items()):
    entry = {name: sum(values) / len(values) for name, values in counters.items()}
    entry["dispatches_sampled"] = max(len(v) for v in counters.values())
    if "FETCH_SIZE" in entry:
        entry["hbm_fetch_bytes_raw"] = entry["FETCH_SIZE"] * 1024
        entry["hbm_fetch_bytes_x2_wide_stream_correction"] = entry["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in entry:
        entry["hbm_write_bytes_raw"] = entry["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in entry and "SQ_BUSY_CYCLES" in entry:
        entry["valu_wave_insts"] = entry["SQ_INSTS_VALU"]
    if "SQ_LDS_BANK_CONFLICT" in entry and entry.get("SQ_LDS_IDX_ACTIVE"):
        entry["lds_conflict_fraction"] = entry["SQ_LDS_BANK_CONFLICT"] / entry["SQ_LDS_IDX_ACTIVE"]
    if entry.get("TCC_HIT_sum") is not None and entry.get("TCC_MISS_sum") is not None:
        total = entry["TCC_HIT_sum"] + entry["TCC_MISS_sum"]
        entry["l2_hit_rate"] = entry["TCC_HIT_sum"] / total if total else None
    summary[kernel] = entry
print(json.dumps(summary, indent=1))
