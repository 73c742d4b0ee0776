#!/usr/bin/env python3
"""pmc_configs.py DIR - folds the per-config rocprofv3 runs of scripts/profile_configs.sh (DIR/cfgN/{stats,pmc_*}) into one
JSON keyed "cfgN:kernel": per-dispatch averages of every counter, the kernel's average duration and share of its config's
kernel time (from the --stats run, NOT from the slower counter runs), and the derived figures bench.py reports:

    hbm_fetch_bytes_raw / hbm_write_bytes_raw   FETCH_SIZE / WRITE_SIZE are in KILOBYTES (MI355X_MICROARCH.md, HBM section);
                                                gfx950 counts wide coalesced streaming reads at half their size - raw values
                                                are kept and labelled as such
    valu_lane_ops_per_second                    SQ_INSTS_VALU x 64 / duration
    wave_wait_inst_fraction / wave_wait_any_fraction   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (ready, the pipe is busy) and SQ_WAIT_ANY /
                                                SQ_WAVE_CYCLES (parked on s_waitcnt or a barrier).  (Round 2 also derived a "VALU busy" fraction
                                                from SQ_ACTIVE_INST_VALU; on gfx950 that counter equals SQ_INSTS_VALU - instructions,
                                                not cycles - so the figure was instructions x an assumed 4 cycles and read above 1.
                                                It is gone; the raw counter stays.)
    lds_conflict_fraction                       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
summary = {}


def short_name(name):
    return name.split("(")[0].replace("void ", "").replace("szs_hip::", "").strip()


for directory in sorted(glob.glob(os.path.join(root, "cfg*"))):
    config = int(re.sub(r"\D", "", os.path.basename(directory)))
    durations, total = {}, 0.0
    for path in glob.glob(os.path.join(directory, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                if "szs_hip" not in row["Name"]:
                    continue
                name = short_name(row["Name"])
                durations[name] = {"_duration_seconds": float(row["AverageNs"]) * 1e-9, "_calls": int(row["Calls"]),
                                   "_total_seconds": float(row["TotalDurationNs"]) * 1e-9}
                total += float(row["TotalDurationNs"]) * 1e-9
    per_kernel = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(directory, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                name = row.get("Kernel_Name", "")
                if "szs_hip" not in name:
                    continue
                short = short_name(name)
                per_kernel[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                per_kernel[short]["_vgpr"] = [float(row.get("VGPR_Count", 0) or 0)]
                per_kernel[short]["_lds"] = [float(row.get("LDS_Block_Size", 0) or 0)]
                per_kernel[short]["_grid"].append(float(row.get("Grid_Size", 0) or 0))
    for kernel in sorted(set(durations) | set(per_kernel)):
        counters = per_kernel.get(kernel, {})
        entry = {name: sum(values) / len(values) for name, values in counters.items()}
        entry["_config"] = config
        entry.update(durations.get(kernel, {}))
        entry["_share"] = entry.get("_total_seconds", 0.0) / total if total else 0.0
        if counters:
            entry["dispatches_sampled"] = max(len(v) for v in counters.values())
        if "FETCH_SIZE" in entry:
            entry["hbm_fetch_bytes_raw"] = entry["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in entry:
            entry["hbm_write_bytes_raw"] = entry["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in entry and entry.get("_duration_seconds"):
            entry["valu_lane_ops_per_second"] = entry["SQ_INSTS_VALU"] * 64 / entry["_duration_seconds"]
        if entry.get("SQ_LDS_IDX_ACTIVE"):
            entry["lds_conflict_fraction"] = entry.get("SQ_LDS_BANK_CONFLICT", 0.0) / entry["SQ_LDS_IDX_ACTIVE"]
        summary[f"cfg{config}:{kernel}"] = entry
    # ---- the whole call: what one C-ABI call of this config issues and moves, summed over its kernels (which may overlap on
    #      several streams: per-kernel durations then add up to more than the call)
    try:
        with open(os.path.join(directory, "bench.json")) as handle:
            line = json.loads(handle.read().strip().splitlines()[-1])
        calls = line["steps"] + line["warmup"]
        call = {"_config": config, "_calls_timed": calls, "_kernel_seconds_per_call": line["roofline"]["kernel_ms"] * 1e-3,
                "_cells_per_call": line["config"]["cells_per_gpu"], "kernels": {}}
        totals = defaultdict(float)
        scoring = {name: entry for name, entry in summary.items()
                   if isinstance(entry, dict) and entry.get("_config") == config and ":__call__" not in name and
                   not any(helper in name for helper in ("plan_kernel", "utf8_transcode", "byte_presence", "alphabet_"))}  # not scoring launches
        launches_per_call = line["roofline"].get("launches_per_step", 1)
        # The run may hold more calls than steps + warm-up (bench.py's fresh-batch leg), and its FIRST call may be another kind of
        # call altogether (codepoints: no alphabet yet, so the per-width launches instead of the one queue launch): the kernels of
        # the steady call are the ones launched at least half as often as the most frequent one; the launches one steady call
        # makes (the bench line's `launches_per_step`) are dealt over them in proportion.
        most = max([entry.get("_calls", 0) for entry in scoring.values()] + [1])
        # (... or at least once per timed call: config 2's run times 400 calls on the same tapes beside its 55 fresh batches)
        steady = {name: entry.get("_calls", 0) / most for name, entry in scoring.items()
                  if entry.get("_calls", 0) * 2 >= most or entry.get("_calls", 0) >= calls}
        weight = sum(steady.values()) or 1.0
        for name, entry in scoring.items():
            per_call = steady.get(name, 0.0) / weight * launches_per_call
            call["kernels"][name.split(":", 1)[1]] = {"launches_per_call": round(per_call, 3), "share_of_kernel_time": round(entry.get("_share", 0.0), 4)}
            if name not in steady:
                call["kernels"][name.split(":", 1)[1]]["only_in_the_cold_call"] = True
            for counter in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                            "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES",
                            "hbm_fetch_bytes_raw", "hbm_write_bytes_raw"):
                if counter in entry:
                    totals[counter] += entry[counter] * per_call
        call.update(totals)
        seconds = call["_kernel_seconds_per_call"]
        if "SQ_INSTS_VALU" in call and seconds:
            call["valu_lane_ops_per_second"] = call["SQ_INSTS_VALU"] * 64 / seconds
            call["valu_lane_ops_per_cell"] = call["SQ_INSTS_VALU"] * 64 / max(call["_cells_per_call"], 1)
        if call.get("SQ_WAVE_CYCLES"):  # where a resident wavefront's cycles go (quad-cycles, disjoint buckets: MI355X_MICROARCH.md)
            call["wave_wait_inst_fraction"] = call.get("SQ_WAIT_INST_ANY", 0.0) / call["SQ_WAVE_CYCLES"]  # ready to issue, pipe busy
            call["wave_wait_any_fraction"] = call.get("SQ_WAIT_ANY", 0.0) / call["SQ_WAVE_CYCLES"]        # parked on s_waitcnt / barrier
            call["wavefronts_per_simd"] = call["SQ_WAVE_CYCLES"] * 4 / 1024 / (seconds * 2.4e9)
        if call.get("SQ_LDS_IDX_ACTIVE"):
            call["lds_conflict_fraction"] = call.get("SQ_LDS_BANK_CONFLICT", 0.0) / call["SQ_LDS_IDX_ACTIVE"]
            call["lds_busy_fraction"] = call["SQ_LDS_IDX_ACTIVE"] / 256 / (seconds * 2.4e9)  # cycles summed over 256 CUs
        summary[f"cfg{config}:__call__"] = call
        # ---- a config whose run holds calls of TWO kinds (config 2: bench.py times the same tapes again - the plain launch behind the
        #      guard - and then a stream of fresh batches - the launch that plans itself) also gets one record per kind: the counters
        #      of the ONE kernel that kind of call launches, over that kernel's own average duration.  bench.py joins its timed leg to
        #      the record of its kind (`roofline.pmc_leg`); the blend above stays for what it is, the run's average call.
        fused = [name for name in steady if "fused" in name]
        plain = [name for name in steady if "fused" not in name]
        if fused and plain:
            for leg, members in (("fresh", fused), ("same_tapes", plain)):
                main = max(members, key=lambda name: scoring[name].get("_share", 0.0))
                entry = scoring[main]
                one = {"_config": config, "_leg": leg, "_kernel_seconds_per_call": entry.get("_duration_seconds", seconds),
                       "_cells_per_call": call["_cells_per_call"],
                       "kernels": {main.split(":", 1)[1]: {"launches_per_call": 1.0, "share_of_kernel_time": 1.0}}}
                for counter in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES",
                                "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "hbm_fetch_bytes_raw", "hbm_write_bytes_raw"):
                    if counter in entry:
                        one[counter] = entry[counter]
                took = one["_kernel_seconds_per_call"]
                if "SQ_INSTS_VALU" in one and took:
                    one["valu_lane_ops_per_second"] = one["SQ_INSTS_VALU"] * 64 / took
                    one["valu_lane_ops_per_cell"] = one["SQ_INSTS_VALU"] * 64 / max(one["_cells_per_call"], 1)
                if one.get("SQ_WAVE_CYCLES"):
                    one["wave_wait_inst_fraction"] = one.get("SQ_WAIT_INST_ANY", 0.0) / one["SQ_WAVE_CYCLES"]
                    one["wave_wait_any_fraction"] = one.get("SQ_WAIT_ANY", 0.0) / one["SQ_WAVE_CYCLES"]
                    one["wavefronts_per_simd"] = one["SQ_WAVE_CYCLES"] * 4 / 1024 / (took * 2.4e9)
                if one.get("SQ_LDS_IDX_ACTIVE"):
                    one["lds_conflict_fraction"] = one.get("SQ_LDS_BANK_CONFLICT", 0.0) / one["SQ_LDS_IDX_ACTIVE"]
                    one["lds_busy_fraction"] = one["SQ_LDS_IDX_ACTIVE"] / 256 / (took * 2.4e9)
                summary[f"cfg{config}:__call__@{leg}"] = one
    except (OSError, ValueError, KeyError, IndexError) as problem:
        print(f"cfg{config}: no call-level record ({problem})", file=sys.stderr)
# the digest of the library these counters were taken on: bench.py joins the instruction counts to ITS run's kernel times and
# flags them stale when the code has changed since (`roofline.pmc_stale`)
try:
    import hashlib
    library = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stringzilla_amd", "lib", "libstringzillas_rocm_shared.so")
    with open(library, "rb") as handle:
        summary["_library_sha256"] = hashlib.sha256(handle.read()).hexdigest()
except OSError as problem:
    print(f"no library digest ({problem})", file=sys.stderr)
print(json.dumps(summary, indent=1))
