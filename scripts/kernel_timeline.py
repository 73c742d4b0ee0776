"""Timeline of the LAST call in a rocprofv3 --kernel-trace CSV: every kernel's start offset and duration (µs).

usage: kernel_timeline.py kernel_trace.csv [gap_us]
Calls are separated by gaps of more than `gap_us` (default 100) with no kernel running.
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
spans = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 1) or 1),
                r.get("Queue_Id", "?")) for r in rows)
calls, current, horizon = [], [], None
for span in spans:
    if horizon is not None and span[0] - horizon > gap:
        calls.append(current)
        current = []
    current.append(span)
    horizon = max(horizon or 0, span[1])
calls.append(current)
last = calls[-1]
origin = last[0][0]
print(f"{len(calls)} calls; last one: {len(last)} kernels over {(max(s[1] for s in last) - origin) / 1e3:.1f} us")
for start, end, name, grid, block, queue in last:
    short = name.split("(")[0].replace("stringzilla_amd::", "")[-70:]
    print(f"  +{(start - origin) / 1e3:9.1f} us  {(end - start) / 1e3:9.1f} us  q{queue:>3}  wg {grid // max(block, 1):6d} x {block:4d}  {short}")
