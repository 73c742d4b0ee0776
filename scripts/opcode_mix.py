#!/usr/bin/env python3
"""opcode_mix.py [OUT.json] - the VALU opcode histogram of every scoring kernel's MAIN LOOP, and the issue ceiling it implies.

The kernels of this path are bound by integer VALU issue.  gfx950's SIMDs are 32 lanes wide: a
wavefront's VALU instruction of the FULL-RATE class (32-bit add / sub / logic / right shift / move) occupies its SIMD for 2
cycles, one of the HALF-RATE class (every maximum, packed 16-bit arithmetic, carries, left shifts and funnel shifts, three-
operand VOP3 forms, DPP moves: scripts/valu_peak.hip measured them all at half the rate) for 4.  A loop of F full-rate and H
half-rate instructions can therefore issue at most

    ceiling = (F + H) / (F / 78.6 + H / 39.3)   x 10^12 lane-operations per second on one MI355X
            (78.6 T = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz, /opt/skills/guides/MI355X_MICROARCH.md "Wave scheduling")

and `bench.py` prices what the PMC passes counted (SQ_INSTS_VALU x 64 / kernel time) against that - a fraction that cannot
exceed 1, unlike round 2's flat 39.3 T.  The histogram comes from the code object's own assembly: this script re-runs hipcc
with the flags of csrc/Makefile and `-S`, takes every `__global__` function, groups its basic blocks by the innermost loop
LLVM's loop comments assign them to, and keeps the loop that holds the most VALU instructions (the kernels' column / step
loops are unrolled, so that IS the main loop; its label is recorded so the choice can be checked by eye).

Which class an opcode belongs to is MEASURED (profiles/r02/valu_peak.json, profiles/r03/team_ops.json: above 50 T -> full).
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "stringzilla_amd", "csrc", "hip")
FULL_RATE_T, HALF_RATE_T = 78.6, 39.3
# full-rate opcodes as measured; every other VALU opcode is priced at half rate (the conservative side for the ceiling's use:
# an unknown opcode lowers the ceiling, which RAISES the reported fraction - so unknowns are listed in the output)
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_bitop3_b32", "v_lshrrev_b32",
             "v_mov_b32", "v_add_f32", "v_xnor_b32", "v_ashrrev_i32"}
MEASURED_HALF = {"v_max_i32", "v_max_u32", "v_min_i32", "v_min_u32", "v_max3_i32", "v_med3_i32", "v_pk_max_u16", "v_pk_max_i16", "v_pk_min_u16",
                 "v_pk_add_i16", "v_pk_add_u16", "v_pk_sub_u16", "v_pk_max_f16", "v_pk_maximum3_f16", "v_alignbit_b32", "v_alignbyte_b32",
                 "v_addc_co_u32", "v_add_co_u32", "v_add3_u32", "v_lshlrev_b32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_bfe_i32",
                 "v_perm_b32", "v_bcnt_u32_b32", "v_cndmask_b32", "v_mad_u32_u24", "v_mul_lo_u32", "v_max_f32", "v_max3_f32", "v_lshl_add_u32",
                 "v_mov_b32_dpp", "v_add_u32_dpp", "v_add_u32_sdwa", "v_lshlrev_b32_sdwa", "v_subb_co_u32", "v_sub_co_u32"}


def kernels_of(source):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-S", "--cuda-device-only"]
    text = subprocess.run(["/opt/rocm/bin/hipcc", *flags, os.path.join(HIP, source), "-o", "-"], check=True, capture_output=True, text=True).stdout
    current, body = None, []
    for line in text.splitlines():
        start = re.match(r"^(_Z\w+):", line)
        if start and current is None:
            current, body = start.group(1), []
        elif current is not None:
            if line.startswith(".Lfunc_end"):
                yield current, body
                current = None
            else:
                body.append(line)


def opcode_of(line):
    token = line.strip().split(None, 1)[0] if line.strip() else ""
    if not token.startswith("v_"):
        return None
    token = re.sub(r"_e(32|64)$", "", token)
    if " row_" in line or " quad_perm" in line or " wave_" in line:
        token += "_dpp" if not token.endswith("_dpp") else ""
    return token


def main_loop(body, prefer_dpp=False):
    """Blocks -> innermost loop header; returns (header label, Counter of VALU opcodes) of the loop with the most VALU."""
    loops, header_of_block, block = defaultdict(Counter), None, None
    for line in body:
        label = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", line)
        if label:
            block, comment = label.group(1), label.group(2) or ""
            inside = re.search(r"in Loop: Header=(BB\d+_\d+)", comment)
            header_of_block = "." + "L" + inside.group(1) if inside else (block if "Loop Header" in comment else None)
            continue
        if re.match(r"^\s*;\s*=>.*Loop Header", line) and block:  # a header whose comment continues on the next lines
            header_of_block = block
            continue
        opcode = opcode_of(line)
        if opcode and header_of_block:
            loops[header_of_block][opcode] += 1
    if not loops:
        return None, Counter()
    # The team tier's main loop exists once per chunk count of a pass since round 4 (weighted_teams.hip: `walk`), so the loop
    # with the most instructions may be the PROFILE BUILD of a rich alphabet instead: among loops that hand values from lane to
    # lane (`row_shr` DPP moves - only the step loops do) the largest one is the whole-pass main loop.
    # (round 6: the fill phase walks its predicated steps four at a time, a loop larger than the main loop - and full of length
    # checks: the main loop is the largest one that hands values over by DPP and compares next to nothing)
    stepping = {name: mix for name, mix in loops.items() if any(opcode.endswith("_dpp") for opcode in mix) and
                sum(count for opcode, count in mix.items() if opcode.startswith("v_cmp")) <= 4} if prefer_dpp else {}
    pool = stepping or loops
    header = max(pool, key=lambda name: sum(pool[name].values()))
    return header, pool[header]


def main():
    out = {"_formula": "ceiling = (F + H) / (F / 78.6 + H / 39.3) T lane-operations/s; F, H = full- / half-rate VALU instructions of the main loop",
           "_full_rate_opcodes": sorted(FULL_RATE)}
    for source in sorted(os.listdir(HIP)):
        if not source.endswith(".hip"):
            continue
        for mangled, body in kernels_of(source):
            pretty = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            short = pretty.split("(")[0].replace("void ", "").replace("szs_hip::", "").strip()
            header, mix = main_loop(body, prefer_dpp="weighted_team_kernel" in pretty)
            total = sum(mix.values())
            if not total:
                continue
            full = sum(count for opcode, count in mix.items() if opcode in FULL_RATE)
            half = total - full
            unknown = sorted(opcode for opcode in mix if opcode not in FULL_RATE and opcode not in MEASURED_HALF)
            out[short] = {"source": source, "main_loop": header, "valu_instructions": total, "full_rate": full, "half_rate": half,
                          "ceiling_Tlane_ops_per_s": round(total / (full / FULL_RATE_T + half / HALF_RATE_T), 2),
                          "opcodes": dict(mix.most_common()), "priced_half_rate_unmeasured": unknown}
    # The one-launch kernel (hip/myers_queue.hip) spends its time in `noinline` BODIES - functions of their own in the code
    # object - while the largest loop of the `__global__` function itself is the table build: its mix is the sum of its
    # bodies' main loops (their ceilings differ by a percent: 59.7 ... 61.6), the kernel's own loop kept beside it.
    for name in [name for name in out if name.startswith("levenshtein_myers_queue_kernel<")]:
        runes = name.split("<", 1)[1].rstrip(">").strip()
        bodies = [body for body in out if (body.startswith("queue_lanes<") and body.rstrip(">").split(",")[-1].strip() == runes) or
                  (body.startswith("queue_team<") and body.split("<", 1)[1].split(",")[1].strip() == runes)]
        if not bodies:
            continue
        full, half = sum(out[body]["full_rate"] for body in bodies), sum(out[body]["half_rate"] for body in bodies)
        out[name] = {"source": out[name]["source"], "main_loop": f"the main loops of its {len(bodies)} bodies (queue_lanes / queue_team), summed",
                     "valu_instructions": full + half, "full_rate": full, "half_rate": half,
                     "ceiling_Tlane_ops_per_s": round((full + half) / (full / FULL_RATE_T + half / HALF_RATE_T), 2),
                     "bodies": sorted(bodies), "table_build_loop": out[name], "priced_half_rate_unmeasured": []}
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as handle:
            handle.write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
