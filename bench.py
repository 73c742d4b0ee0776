#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's configs, through the C-ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2] [--extra-configs 3,4,5,6] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric : DP cell-updates/s (GCUPS) = sum over scored pairs of len(query) * len(candidate) / seconds / 1e9
         (the reference's accounting, /root/reference/bench/similarities.cuh:344-366).
Step   : one `szs_levenshtein_distances_u32tape` call - the whole synchronous C-ABI call, host planning included - over
         config 2: a 1024 x 1024 cross-product (1,048,576 pairs) of printable-ASCII strings, length U[96,160], unit costs.
         Tapes and the results matrix are resident in HBM before the timed region starts.
N > 1  : one process per GPU.  The HEADLINE shards by QUERY ROW BLOCKS: every rank scores its own
         1024 query rows against the same 1024 candidates (broadcast once over RCCL/xGMI before timing), so per-GPU work
         is fixed: "weak" scaling, and the timed path has no collective (rows are independent; results stay sharded).
Line   : rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM with ALGORITHMIC bytes
         (272 B per pair at len 128: len(q) + len(c) + 2 offsets + one 8-byte result) over the
         hipEvent-measured kernel time the library records on its own stream.  `cpu_baseline` times the reference's own
         SIMD engines (oracle/_ref, built from /root/reference) on this box's host cores - a reported baseline only.
configs: the same line carries a `configs` array - one record per other BASELINE.json config (9: config 2 at a fixed 128 bytes,
         its "peak" variant; 3: NW BLOSUM62, 4: SW
         affine NUC.4.4, 5: byte-level Levenshtein on Zipf UTF-8, 5u: the same at the codepoint level, 7 / 8: config 2's batch
         under non-unit costs, linear 1/3/3 and affine 0/1/4/2), each timed
         through its own C-ABI entry point with its kernel and wall GCUPS, checksum, HBM roofline, the VALU counters of
         its dominant kernel (profiles/r02, committed PMC passes) and the reference's Ice Lake engine as `cpu_baseline`
         on a stated sample whose cells are also compared with the GPU's.  With N > 1 configs 4 and 5 are STRONG-scaled
         the way BASELINE.json specifies them: ONE batch, rows dealt over the ranks by LPT (`stringzilla_amd/sharded.py`),
         per-GPU busy time, imbalance = max / mean, aggregate GCUPS - plus the same batch through the single-process C
         entry `szs_rocm_node_*` (one host thread per GPU) when the library exports it.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# The binding resource of this path is integer VALU issue, not HBM.  Its ceiling is MEASURED, not quoted: the Myers
# column update (the kernel's exact instruction mix) on register-resident match masks, no LDS and no memory, sustains
# this many DP cells per second at full bit-vector width on one MI355X (scripts/valu_peak.hip -> profiles/).
PROFILE_DIRS = [os.path.join(ROOT, "profiles", name) for name in ("r04", "r03", "r02", "r01")]
# VALU issue ceiling, class-weighted: gfx950's SIMDs are 32 lanes wide, a full-rate VALU instruction (32-bit add / sub / logic /
# right shift / move) takes a wavefront 2 cycles, every other one (maxima, packed 16-bit, carries, funnel shifts, VOP3 three-
# operand forms, DPP) 4 - MI355X_MICROARCH.md "Wave scheduling", measured per opcode by scripts/valu_peak.hip.  A main loop of F
# full-rate and H half-rate instructions issues at most (F + H) / (F / 78.6 + H / 39.3) T lane-operations/s; F and H come from
# the code object's own assembly (scripts/opcode_mix.py -> profiles/r03/opcode_mix.json).
VALU_FULL_RATE_PEAK, VALU_HALF_RATE_PEAK = 78.6e12, 39.3e12  # 256 CUs x 4 SIMDs x 32 (16) lanes x 2.4 GHz


def _profile_json(name):
    for directory in PROFILE_DIRS:
        try:
            with open(os.path.join(directory, name)) as handle:
                return json.load(handle), os.path.relpath(os.path.join(directory, name), ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--config", type=int, default=2, help="BASELINE.json config index of the headline (2 = the metric's config)")
    parser.add_argument("--generator", default="mt19937_64", choices=["numpy", "mt19937_64"],
                        help="where configs 1-4 come from: std::mt19937_64, the generator SURVEY.md section 8(d) names "
                             "(tests/native/workloads_mt19937.cpp spells the mapping out; reproducible from C++), or numpy's "
                             "default_rng (the batches rounds 1-3 were profiled on: the same shapes, other strings).  Configs 5 / 5u "
                             "(Zipf UTF-8) exist in numpy only.  Falls back to numpy, and says so, when the helper library is not built")
    parser.add_argument("--extra-configs", default=None,
                        help="comma-separated configs reported in the `configs` array (default: 3,4,5,6,7,8 on one GPU, "
                             "4,5 strong-scaled on several; 'none' to skip)")
    parser.add_argument("--extra-seconds", type=float, default=4.0, help="GPU time budget per extra config")
    parser.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget per cpu_baseline sample")
    parser.add_argument("--extra-scale", type=float, default=1.0,
                        help="testing aid: shrinks the matrix side of the `configs` records (1.0 = BASELINE.json's sizes)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    parser.add_argument("--same-device", action="store_true",
                        help="testing aid: every rank uses cuda:0 (with --backend gloo), to exercise the N > 1 code "
                             "path on a one-GPU box; the numbers of such a run mean nothing")
    parser.add_argument("--hbm-traffic-bytes", type=float, default=None,
                        help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: the committed summary "
                             "profiles/rNN/pmc_summary.json of this same command (scripts/profile_gpu.sh)")
    return parser.parse_args()


def resolve_generator(wanted):
    """`mt19937_64` needs tests/native/bin/libworkloads_mt19937.so (built by __graft_entry__.build()); without it: numpy."""
    if wanted == "mt19937_64" and not os.path.exists(os.path.join(ROOT, "tests", "native", "bin", "libworkloads_mt19937.so")):
        return "numpy"
    return wanted


def fresh_tape(generator, seed, count, low, high):
    """Another batch of printable ASCII from the run's generator (the headline alternates two batches)."""
    from stringzilla_amd import workloads

    if generator == "mt19937_64":
        return workloads.mt19937_64_tape(seed, count, low, high, workloads.ASCII_PRINTABLE)
    return workloads.random_tape(np.random.default_rng(seed), count, low, high, workloads.ASCII_PRINTABLE)


def library_digest():
    """sha256 of the code the run executes: committed PMC passes carry the digest of the library they counted (scripts/
    pmc_configs.py); when it differs, instruction counts joined to this run's kernel times are flagged stale."""
    import hashlib

    from stringzilla_amd import _abi

    with open(_abi.LIBRARY_PATH, "rb") as handle:
        return hashlib.sha256(handle.read()).hexdigest()


def measured_hbm_peak(where):
    """What this box's HBM delivers to a plain device-to-device copy of 1 GiB (read + write bytes over the time of the copy,
    best of five): the measured peak beside the 8 TB/s of the data sheet (SURVEY.md section 8d asks for both)."""
    import torch

    size = 1 << 30
    source = torch.empty(size, dtype=torch.uint8, device=where)
    target = torch.empty(size, dtype=torch.uint8, device=where)
    source.fill_(1)
    target.copy_(source)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        target.copy_(source)
        end.record()
        torch.cuda.synchronize()
        seconds = begin.elapsed_time(end) * 1e-3
        best = seconds if best is None else min(best, seconds)
    del source, target
    return round(2.0 * size / best / 1e9, 1)


# ---- engines and entry points per workload kind -------------------------------------------------------------------------

ENTRY_POINTS = {
    "levenshtein": "szs_levenshtein_distances_u32tape", "levenshtein_utf8": "szs_levenshtein_distances_utf8_u32tape",
    "needleman_wunsch": "szs_needleman_wunsch_scores_u32tape", "smith_waterman": "szs_smith_waterman_scores_u32tape",
}


def make_engine(load, scope):
    import stringzilla_amd as szs
    from stringzilla_amd import matrices

    if load.kind == "levenshtein":
        return szs.LevenshteinDistances(**load.costs, capabilities=scope)
    if load.kind == "levenshtein_utf8":
        return szs.LevenshteinDistancesUTF8(**load.costs, capabilities=scope)
    cls = szs.NeedlemanWunschScores if load.kind == "needleman_wunsch" else szs.SmithWatermanScores
    return cls(*matrices.by_name(load.table), **load.costs, capabilities=scope)


def make_step(engine, scope, load, queries, candidates, results, device_index):
    """The raw C-ABI call of this workload as a closure (ctypes only inside the timed loop)."""
    from stringzilla_amd import _abi

    call = getattr(_abi.lib, ENTRY_POINTS[load.kind])
    q_tape, c_tape = queries._tape(device_index), candidates._tape(device_index)
    error = ctypes.c_char_p()
    columns = len(candidates)

    def step():
        status = call(engine.handle, scope.handle, ctypes.byref(q_tape), ctypes.byref(c_tape), results.data_ptr(), columns,
                      ctypes.byref(error))
        if status:
            raise RuntimeError(f"{ENTRY_POINTS[load.kind]} failed: {status} {error.value}")

    step.keepalive = (q_tape, c_tape, queries, candidates)
    return step


# ---- the reference's CPU engines beside it ---------------------------------------------------------------------------------

def cpu_baseline(load, gpu_matrix, seconds):
    """Times the reference's own CPU engines (best SIMD tier, all host threads) on a BOUNDED sample of the same batch -
    evenly spaced query rows x evenly spaced candidates, sized by a calibration pass to about `seconds` of CPU work, the
    whole batch when that fits - and checks that they produce the very cells the GPU produced.  Test-infrastructure
    code path: the only place bench.py touches oracle/."""
    from oracle import binding
    from stringzilla_amd import matrices

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    checker, kind, label = None, None, None
    if binding.reference_available():
        try:
            tier = binding.reference_best_tier()
            checker = binding.reference(tier=tier, threads=cores)
            kind, label = "reference", {0: "serial", 1: "haswell (AVX2)", 2: "icelake (AVX-512)"}[tier]
        except OSError:
            checker = None
    if checker is None:
        checker, kind, label, cores = binding.oracle(), "port", "plain-C oracle", 1

    def run(rows, columns):
        queries = [load.queries[int(i)] for i in rows]
        candidates = [load.candidates[int(j)] for j in columns]
        if load.kind == "levenshtein":
            return checker.levenshtein(queries, candidates, **load.costs)
        if load.kind == "levenshtein_utf8":
            return checker.levenshtein_utf8(queries, candidates, **load.costs)
        scorer = checker.needleman_wunsch if load.kind == "needleman_wunsch" else checker.smith_waterman
        return scorer(queries, candidates, *matrices.by_name(load.table), **load.costs)

    q_lengths, c_lengths = load.queries.lengths(), load.candidates.lengths()
    spaced = lambda count, take: np.unique(np.linspace(0, count - 1, num=max(1, min(count, take))).round().astype(np.int64))
    cells_of = lambda rows, columns: float(q_lengths[rows].sum()) * float(c_lengths[columns].sum())

    # calibration: one row per thread (the shim deals contiguous row blocks to threads) x 64 candidates - the reference's SIMD
    # engines score one query against 16 / 32 / 64 candidates at once, one per lane, so a sample's candidates come in whole
    # multiples of 64 (round 2's config-4 sample of 26 candidates left a third of the lanes empty and read 26 GCUPS for ~32)
    lanes = 64
    rows = spaced(len(q_lengths), max(cores, 1))
    columns = spaced(len(c_lengths), lanes)
    started = time.perf_counter()
    run(rows, columns)
    rate = cells_of(rows, columns) / max(time.perf_counter() - started, 1e-4)
    budget_cells = rate * seconds / 5  # one run of the sample takes about a fifth of the budget: five or more timed repeats
    # grow the sample towards the budget: first more candidates (whole lane groups), then more rows
    per_column = cells_of(rows, np.arange(len(c_lengths))) / len(c_lengths)
    take_columns = int(min(len(c_lengths), max(lanes, budget_cells / max(per_column, 1.0) // lanes * lanes)))
    columns = spaced(len(c_lengths), take_columns)
    if take_columns == len(c_lengths):
        per_row = cells_of(np.arange(len(q_lengths)), columns) / len(q_lengths)
        rows = spaced(len(q_lengths), int(min(len(q_lengths), max(len(rows), budget_cells / max(per_row, 1.0)))))
    whole = len(rows) == len(q_lengths) and len(columns) == len(c_lengths)

    started = time.perf_counter()
    matrix = run(rows, columns)
    first = time.perf_counter() - started
    expected = gpu_matrix[np.ix_(rows, columns)]
    assert np.array_equal(matrix.view(np.int64), expected.view(np.int64)), "CPU baseline and GPU disagree"
    # five or more runs when the budget allows; a batch whose SMALLEST fair sample (a row per thread x one lane group) already
    # takes seconds per run - config 4 - gets three
    # ... and a run lasts a quarter of a second or more: a batch the CPU finishes in 30 ms (configs 2, 7, 8) is passed several
    # times back to back per run - timed one pass at a time, 256 threads starting up made the median wander between 340 and 580
    passes = 1 if first >= 0.25 else int(min(64, np.ceil(0.25 / max(first, 1e-4))))
    repeats = int(max(3 if first * passes > seconds / 5 else 5, min(20, (seconds - first) / max(first * passes, 1e-3))))
    runs = []
    for _ in range(repeats):
        started = time.perf_counter()
        for _ in range(passes):
            run(rows, columns)
        runs.append((time.perf_counter() - started) / passes)
    elapsed = float(np.median(runs))  # the median: a 256-thread host shows stragglers, a mean of two runs wandered by 25 %
    # the reference reports its Serial engine beside the SIMD tiers (similarities/README.md:21-24): ONE thread of the serial tier
    # on a sample of about a second (a row per eight of the SIMD sample's, one lane group of candidates)
    serial = None
    if kind == "reference":
        try:
            plain = binding.reference(tier=0, threads=1)
            serial_rows, serial_columns = spaced(len(q_lengths), 4), spaced(len(c_lengths), lanes)
            def run_serial():
                queries = [load.queries[int(i)] for i in serial_rows]
                candidates = [load.candidates[int(j)] for j in serial_columns]
                if load.kind == "levenshtein":
                    return plain.levenshtein(queries, candidates, **load.costs)
                if load.kind == "levenshtein_utf8":
                    return plain.levenshtein_utf8(queries, candidates, **load.costs)
                scorer = plain.needleman_wunsch if load.kind == "needleman_wunsch" else plain.smith_waterman
                return scorer(queries, candidates, *matrices.by_name(load.table), **load.costs)
            started = time.perf_counter()
            got = run_serial()
            once = time.perf_counter() - started
            assert np.array_equal(got.view(np.int64), gpu_matrix[np.ix_(serial_rows, serial_columns)].view(np.int64)), "serial CPU baseline and GPU disagree"
            again = int(max(1, min(50, 1.0 / max(once, 1e-4))))
            started = time.perf_counter()
            for _ in range(again):
                run_serial()
            serial = {"value": round(cells_of(serial_rows, serial_columns) * again / (time.perf_counter() - started) / 1e9, 3), "unit": "GCUPS",
                      "cores": 1, "sample": f"{len(serial_rows)} query rows x {len(serial_columns)} candidates, {again} passes, serial tier, cells verified equal"}
        except AssertionError:
            raise
        except Exception as problem:
            serial = {"error": repr(problem)}
    what = (f"the full {len(q_lengths)}x{len(c_lengths)} batch of the timed config" if whole else
            f"{len(rows)} evenly spaced query rows x {len(columns)} evenly spaced candidates of the timed config")
    return {
        "value": round(cells_of(rows, columns) / elapsed / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": kind,
        "spread": [round(cells_of(rows, columns) / max(runs) / 1e9, 2), round(cells_of(rows, columns) / min(runs) / 1e9, 2)],
        "sample": f"{what}, median of {repeats} runs{f' of {passes} passes each' if passes > 1 else ''} (`spread`: slowest and fastest run), {label} tier, {cores} threads, tape packing "
                  f"included; the sampled cells verified equal to the GPU's",
        "serial_1_thread": serial,
    }


# ---- rooflines -------------------------------------------------------------------------------------------------------------

def roofline(config, profile, kernel_seconds, traffic_override=None):
    """HBM roofline from the ALGORITHMIC bytes of one call over the kernel time measured live (hipEvent pair on the library's
    stream), beside what the committed rocprofv3 --pmc passes of this same command saw: HBM bytes actually moved per call
    and - what actually binds this path - VALU issue and LDS occupancy (scripts/profile_configs.sh -> profiles/rNN)."""
    achieved = profile.algorithmic_bytes / kernel_seconds / 1e9
    record = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic_override, "kernel_ms": round(kernel_seconds * 1e3, 4),
              "algorithmic_bytes": int(profile.algorithmic_bytes), "launches_per_step": int(profile.launches),
              "kernel_gcups": round(profile.cells / kernel_seconds / 1e9, 1)}
    summary, where = _profile_json("pmc_configs.json")  # "cfgN:kernel" and "cfgN:__call__" (scripts/pmc_configs.py)
    call = (summary or {}).get(f"cfg{config}:__call__")
    counted_on = (summary or {}).get("_library_sha256")
    record["pmc_library_sha256"] = counted_on
    record["pmc_stale"] = None if not counted_on else counted_on != library_digest()  # True: the counters below were taken on other code
    if not call:
        record["traffic_source"] = "no committed PMC pass for this config" if traffic_override is None else "from --hbm-traffic-bytes"
        return record
    kernels = call.get("kernels", {})
    if kernels:
        dominant = max(kernels, key=lambda name: kernels[name]["share_of_kernel_time"])
        record["kernel"] = dominant if len(kernels) == 1 else f"{dominant} + {len(kernels) - 1} more (launches of different widths overlap on {int(getattr(profile, 'streams', 0)) or 'several'} streams)"
    if traffic_override is None and "hbm_fetch_bytes_raw" in call and "hbm_write_bytes_raw" in call:
        record["traffic"] = round(call["hbm_fetch_bytes_raw"] + call["hbm_write_bytes_raw"])
        record["traffic_source"] = (f"{where}: (FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernels of one call, committed rocprofv3 "
                                    f"--pmc passes of this command (raw; wide-stream reads may count double)")
    if "SQ_INSTS_VALU" in call:
        lane_ops = call["SQ_INSTS_VALU"] * 64.0
        mixes, mix_where = _profile_json("opcode_mix.json")
        # the ceiling of the call = its kernels' ceilings weighted by their share of the kernel time
        weights = {name: kernels[name]["share_of_kernel_time"] for name in kernels if (mixes or {}).get(name)}
        if weights:
            total = sum(weights.values())
            ceiling = 1e12 * total / sum(share / mixes[name]["ceiling_Tlane_ops_per_s"] for name, share in weights.items())
            main = max(weights, key=weights.get)
            mix = {"kernel": main, "main_loop": mixes[main]["main_loop"], "full_rate_instructions": mixes[main]["full_rate"],
                   "half_rate_instructions": mixes[main]["half_rate"], "source": mix_where}
        else:
            ceiling, mix = VALU_HALF_RATE_PEAK, None
        record["valu"] = {
            "bound": "integer VALU issue (PMC), class-weighted ceiling", "source": where,
            "wave_instructions_per_call": round(call["SQ_INSTS_VALU"]),
            "lane_ops_per_cell": round(lane_ops / max(float(profile.cells), 1.0), 4),
            "achieved_Tlane_ops_per_s": round(lane_ops / kernel_seconds / 1e12, 2),
            "peak_Tlane_ops_per_s": round(ceiling / 1e12, 2),
            "frac": round(lane_ops / kernel_seconds / ceiling, 4),
            "opcode_mix": mix,
            "lds_busy_fraction": round(call["lds_busy_fraction"], 4) if "lds_busy_fraction" in call else None,
            "lds_conflict_fraction": round(call["lds_conflict_fraction"], 4) if "lds_conflict_fraction" in call else None,
            "wave_cycles_waiting_to_issue": round(call["wave_wait_inst_fraction"], 4) if "wave_wait_inst_fraction" in call else None,
            "wave_cycles_parked": round(call["wave_wait_any_fraction"], 4) if "wave_wait_any_fraction" in call else None,
            "note": "frac = (VALU wave-instructions of one call x 64 lanes / live kernel seconds) over (F + H) / (F / 78.6 T + H / 39.3 T), "
                    "F / H = full- / half-rate instructions of the dominant kernels' main loops (time-weighted over the kernels of the "
                    "call); the instruction count is a committed PMC pass of this command, the kernel time is this run's",
        }
    return record


def time_config(step, engine, budget_seconds, fence, floor=2, ceiling=50):
    """Warm-up call, then as many timed calls as `budget_seconds` allows; returns (wall s per call, kernel s per call)."""
    step()
    fence()
    started = time.perf_counter()
    step()
    once = time.perf_counter() - started
    repeats = int(max(floor, min(ceiling, budget_seconds / max(once, 1e-4))))
    kernel = []
    fence()
    started = time.perf_counter()
    for _ in range(repeats):
        step()
        kernel.append(engine.last_call_profile().kernel_milliseconds * 1e-3)
    fence()
    return (time.perf_counter() - started) / repeats, float(np.mean(kernel)), repeats


def measure_extra(config, scope, device_index, args, fence, with_cpu):
    """One record of the `configs` array on ONE GPU: the whole batch through the config's own entry point."""
    import torch

    from stringzilla_amd import workloads

    load = workloads.config(config, scale=args.extra_scale, generator=args.generator if config <= 4 or config == 9 else "numpy")
    engine = make_engine(load, scope)
    queries, candidates = load.queries.to_device(device_index), load.candidates.to_device(device_index)
    results = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device=torch.device("cuda", device_index))
    step = make_step(engine, scope, load, queries, candidates, results, device_index)
    wall, kernel, repeats = time_config(step, engine, args.extra_seconds, fence)
    profile = engine.last_call_profile()
    record = {
        "config": config, "workload": load.name, "entry_point": ENTRY_POINTS[load.kind], "n_gpus": 1,
        "pairs": int(profile.pairs), "cells": int(profile.cells), "steps": repeats,
        "value": round(profile.cells / wall / 1e9, 1), "unit": "GCUPS", "ms_per_step": round(wall * 1e3, 3),
        "kernel_gcups": round(profile.cells / kernel / 1e9, 1), "tier": int(profile.tier), "cell_bits": int(profile.cell_bits),
        "dtype": {0: "u32 bit-vectors", 16: "i16 cells, two per VALU op", 32: "i32 cells", 64: "i64 cells"}.get(int(profile.cell_bits), "?"),
        "results_checksum": int(results.sum().item()),
        "roofline": roofline(config, profile, kernel),
    }
    if with_cpu:  # timed later, after every GPU measurement of the run (the host cores are busy for ~10 s per baseline); the
        # matrix stays in HBM until then - downloading 80 MB here would idle the shader engines right before the headline
        record["_cpu_baseline_inputs"] = (load, results)
    return record


def measure_fingerprints(scope, device_index, args, fence):
    """`szs_fingerprints_u32tape`: rolling MinHash over 1024 documents of ~10 KB, 1024 dimensions of the
    reference's default window widths - bytes of text per second and byte-dimensions per second of the whole C-ABI call.  The
    kernel's work is 25 instructions per byte and dimension: VALU-bound, the text is read once per 256
    dimensions from LDS."""
    import torch

    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    dimensions = 1024
    texts = workloads.random_tape(np.random.default_rng(11), 1024, 8192, 12288, workloads.ASCII_PRINTABLE).to_device(device_index)
    engine = szs.Fingerprints(dimensions, capabilities=scope)
    engine(texts, device=scope)
    fence()
    started = time.perf_counter()
    engine(texts, device=scope)
    once = time.perf_counter() - started
    repeats = int(max(3, min(50, args.extra_seconds / max(once, 1e-4))))
    # two passes, the faster one counts: the first pass after an engine is created has been seen at twice the time of every later
    # one (12.7 ms, then 6.2 ms and staying there: profiles/r04/fingerprints_first_pass.txt) - grow-only buffers and clocks settling
    walls = []
    for _ in range(2):
        fence()
        started = time.perf_counter()
        for _ in range(repeats):
            hashes, counts = engine(texts, device=scope)
        fence()
        walls.append((time.perf_counter() - started) / repeats)
    wall = min(walls)
    text_bytes = int(texts.lengths().sum())
    # Instructions per byte and dimension from the kernel's own assembly: its main loop is unrolled over FOUR positions
    # (`#pragma unroll 4`, hip/fingerprints.hip), so VALU instructions of that loop / 4; the ceiling is that loop's class-weighted
    # one (scripts/opcode_mix.py).  An estimate - no PMC pass counts this call - and said so; round 3 assumed 25 per position
    # against the flat half-rate peak and could read above 1.
    mixes, mix_where = _profile_json("opcode_mix.json")
    mix = (mixes or {}).get("fingerprint_segments_kernel")
    per_position = mix["valu_instructions"] / 4.0 if mix else 25.0
    ceiling = mix["ceiling_Tlane_ops_per_s"] * 1e12 if mix else VALU_HALF_RATE_PEAK
    lane_ops = per_position * text_bytes * dimensions  # one lane per dimension: instructions per position x positions x lanes
    return {"config": "fingerprints", "workload": "1024 ASCII documents of 8-12 KB, 1024 dimensions, default window widths",
            "entry_point": "szs_fingerprints_u32tape", "n_gpus": 1, "steps": repeats, "ms_per_step": round(wall * 1e3, 3),
            "passes_ms": [round(w * 1e3, 3) for w in walls],
            "value": round(text_bytes * dimensions / wall / 1e12, 3), "unit": "10^12 byte-dimensions/s",
            "text_gb_s": round(text_bytes / wall / 1e9, 2), "results_checksum": int(hashes.astype(np.uint64).sum() % (1 << 53)),
            "roofline": {"bound": f"integer / fp64 VALU issue, ESTIMATED: {per_position:.2f} instructions per byte and dimension "
                                  f"(main loop of the kernel's assembly / 4 positions) against that loop's class-weighted ceiling"
                                  f"{' (' + mix_where + ')' if mix else ' (no opcode mix committed: 25 assumed, flat half-rate peak)'}; wall time, no PMC pass",
                         "achieved_Tlane_ops_per_s": round(lane_ops / wall / 1e12, 2), "peak_Tlane_ops_per_s": round(ceiling / 1e12, 2),
                         "frac": round(lane_ops / wall / ceiling, 4),
                         "hbm": {"algorithmic_bytes": text_bytes + 8 * dimensions * len(texts), "achieved_gb_s": round((text_bytes + 8 * dimensions * len(texts)) / wall / 1e9, 2),
                                 "peak": HBM_PEAK_GBPS}}}


def attach_cpu_baselines(records, seconds):
    for record in records:
        inputs = record.pop("_cpu_baseline_inputs", None)
        if inputs is None:
            continue
        try:
            record["cpu_baseline"] = cpu_baseline(inputs[0], inputs[1].cpu().numpy(), seconds)
        except AssertionError:
            raise
        except Exception as problem:  # the checker is optional equipment; the GPU numbers stand without it
            record["cpu_baseline"] = {"error": repr(problem)}


def measure_strong(config, scope, device_index, args, fence, dist, world, rank, where):
    """Configs 4 and 5 as BASELINE.json states them: ONE batch whose query rows are dealt over the GPUs of the node.
    Rank 0 owns the batch; tapes are replicated by an RCCL broadcast (device to device); every rank scores its rows; no
    collective on the data path.  Reports per-GPU busy time, imbalance and the aggregate rate."""
    import torch

    from stringzilla_amd import sharded, workloads

    load = workloads.config(config, scale=args.extra_scale, generator=args.generator if config <= 4 or config == 9 else "numpy")  # seeded: every rank can name the engine; only rank 0's strings are used
    engine = make_engine(load, scope)
    busy, state = [], {}

    def score(queries, candidates):
        queries.to_device(device_index), candidates.to_device(device_index)
        out = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device=where)
        step = make_step(engine, scope, load, queries, candidates, out, device_index)
        wall, kernel, repeats = time_config(step, engine, args.extra_seconds, lambda: torch.cuda.synchronize(), floor=2, ceiling=20)
        busy.append(wall), state.update(kernel=kernel, repeats=repeats)
        return out

    node = sharded.ShardedEngine(engine=engine, scope=scope, score=score)
    fence()
    rows, local = node(load.queries if rank == 0 else None, load.candidates if rank == 0 else None, source=0)
    profile = engine.last_call_profile()
    mine = torch.tensor([busy[0] if busy else 0.0, float(profile.cells) if busy else 0.0, float(len(rows)),
                         float(local.sum()) if len(rows) else 0.0, state.get("kernel", 0.0)], dtype=torch.float64,
                        device=where if dist.get_backend() == "nccl" else "cpu")
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank:
        return None
    seconds = np.array([float(g[0]) for g in gathered])
    cells = sum(float(g[1]) for g in gathered)
    return {
        "config": config, "workload": load.name, "entry_point": ENTRY_POINTS[load.kind], "n_gpus": world, "scaling": "strong",
        "sharding": "query rows dealt by LPT on len(query), both tapes replicated by RCCL broadcast, results stay sharded",
        "cells": int(cells), "rows_per_gpu": [int(g[2]) for g in gathered],
        "busy_ms_per_gpu": [round(s * 1e3, 3) for s in seconds.tolist()],
        "kernel_ms_per_gpu": [round(float(g[4]) * 1e3, 3) for g in gathered],
        "imbalance_max_over_mean": round(float(seconds.max() / max(seconds.mean(), 1e-12)), 4),
        "row_weight_imbalance": round(node.last_balance, 4),
        "value": round(cells / seconds.max() / 1e9, 1), "unit": "GCUPS",
        "results_checksum": int(sum(float(g[3]) for g in gathered)), "same_device": bool(args.same_device),
    }


def measure_c_node(config, devices, args):
    """The same strong-scaled batch through the single-process C entry (`szs_rocm_node_*`): one host thread per GPU inside
    the library, tapes replicated by `hipMemcpyPeerAsync`.  Runs on rank 0 while the other ranks wait at a barrier."""
    import torch

    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    if not hasattr(szs, "Node"):
        return None
    load = workloads.config(config, scale=args.extra_scale, generator=args.generator if config <= 4 or config == 9 else "numpy")
    node = szs.Node(devices)
    engine = node.engine_for(load)
    out = torch.empty((len(load.queries), len(load.candidates)), dtype=torch.int64, device=torch.device("cuda", devices[0]))
    load.queries.to_device(devices[0]), load.candidates.to_device(devices[0])
    engine(load.queries, load.candidates, out=out)
    started = time.perf_counter()
    engine(load.queries, load.candidates, out=out)
    repeats = int(max(5, min(20, args.extra_seconds / max(time.perf_counter() - started, 1e-3))))
    started = time.perf_counter()
    for _ in range(repeats):
        stats = engine(load.queries, load.candidates, out=out)
    wall = (time.perf_counter() - started) / repeats
    return {"config": config, "entry_point": "szs_rocm_node_scores_u32tape", "n_gpus": len(devices), "scaling": "strong", "steps": repeats,
            "value": round(load.cells / wall / 1e9, 1), "unit": "GCUPS", "ms_per_step": round(wall * 1e3, 3),
            "peer_pairs": stats.get("peer_pairs"), "peer_copies": stats.get("peer_copies"), "staged_copies": stats.get("staged_copies"),
            "busy_ms_per_gpu": [round(x, 3) for x in stats["busy_ms"]], "rows_per_gpu": stats["rows"],
            "results_checksum": int(out.sum().item())}


def main():
    args = parse_args()
    # The application's choice, made before HIP initialises: twelve hardware queues, so that the per-width launches of a
    # mixed-length batch (configs 5 / 5u: up to eight streams) each get their own.  The library itself never writes the
    # environment; it reads this variable when it is loaded and sizes its fan-out to it (csrc/host/tuning.c).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    import torch
    import torch.distributed as dist

    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    where = torch.device("cuda", local_rank)

    # ---- the batch: rank r owns query rows [1024 r, 1024 (r + 1)); candidates are shared by all ranks
    args.generator = resolve_generator(args.generator)
    load = workloads.config(args.config, generator=args.generator if args.config <= 4 else "numpy")
    if rank:
        low, high = int(load.queries.lengths().min()), int(load.queries.lengths().max())
        load.queries = fresh_tape(args.generator, args.config + 1000 * rank, len(load.queries), 96 if args.config == 2 else low,
                                  160 if args.config == 2 else high)
    queries = load.queries.to_device(local_rank)
    if world > 1:  # the one exchange step of the path: replicate the candidates tape over RCCL / xGMI, before timing;
        # the received tape stays in HBM (only its offsets are mirrored on the host)
        size = torch.tensor([load.candidates.data.size], device=where)
        dist.broadcast(size, 0)
        data = torch.from_numpy(load.candidates.data).to(where) if rank == 0 else torch.empty(int(size), dtype=torch.uint8, device=where)
        offsets = torch.from_numpy(load.candidates.offsets.view(np.int32)).to(where) if rank == 0 else torch.empty(len(load.candidates) + 1, dtype=torch.int32, device=where)
        dist.broadcast(data, 0)
        dist.broadcast(offsets, 0)
        load.candidates = szs.Strs.from_device(data, offsets)
    candidates = load.candidates.to_device(local_rank)

    scope = szs.DeviceScope(gpu_device=local_rank)
    engine = make_engine(load, scope)
    rows, columns = len(queries), len(candidates)
    results = torch.empty((rows, columns), dtype=torch.int64, device=where)
    step = make_step(engine, scope, load, queries, candidates, results, local_rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the other configs run BEFORE the headline's warm-up: a run of W = 5 short warm-up steps on a GPU that has idled
    # through the set-up is timed on its clock ramp (round 1: 64.4 TCUPS with 20 steps against 67.9 with 200); after seconds
    # of NW / SW scoring the clocks are where a busy GPU keeps them.  The headline itself is unchanged: W untimed steps, then
    # exactly K timed ones.  (The single-process C driver of N > 1 runs last, while the other ranks wait.)
    if args.extra_configs is None:
        extras = [9, 3, 4, 5, 6, 7, 8] if world == 1 else [4, 5]
    else:
        extras = [] if args.extra_configs.strip().lower() in ("", "none") else [int(x) for x in args.extra_configs.split(",")]
    records = []
    for config in extras:
        try:
            if world == 1:
                records.append(measure_extra(config, scope, local_rank, args, fence, not args.no_cpu_baseline))
            else:  # configs 4 and 5 strong-scaled over the ranks; every rank takes part
                record = measure_strong(config, scope, local_rank, args, fence, dist, world, rank, where)
                if record is not None:
                    records.append(record)
        except AssertionError:
            raise
        except Exception as problem:  # an extra record must never cost the headline line
            records.append({"config": config, "error": repr(problem)})

    if world == 1 and args.extra_configs is None:
        try:
            records.append(measure_fingerprints(scope, local_rank, args, fence))
        except Exception as problem:
            records.append({"config": "fingerprints", "error": repr(problem)})

    # ---- the headline is a STREAM OF FRESH BATCHES: two different batches of the config's shape alternate, so that no call
    # finds the plan of its own tapes on the device (csrc/host/dispatch.c re-uses that plan after validating it in the kernels -
    # the best case, which a real stream of batches never meets; round 2's headline measured it).  Every call pays for the
    # planner kernel, speculated launches behind it.  The re-use path is reported beside it as `same_tapes`, measured first.
    # Both legs run after the `configs` records: the first ~50 launches of this kernel after anything else - idling, NW / SW
    # scoring, GEMMs alike - run ~10 % slower (scripts/kernel_ms_series.py: the power management settles on the new instruction
    # mix in ~10 ms), so the W warm-up steps of a short run would otherwise be timed on that transient.
    steps_of, cells_of_step, same_tapes = [step], None, None
    if args.config == 2:  # every rank, on its own GPU
        other = fresh_tape(args.generator, 4242 + 10 * rank, len(load.queries), 96, 160).to_device(local_rank)
        other_candidates = fresh_tape(args.generator, 4243 + 10 * rank, len(load.candidates), 96, 160).to_device(local_rank)
        other_step = make_step(engine, scope, load, other, other_candidates, results, local_rank)
        steps_of.append(other_step)
        for _ in range(max(2, args.warmup // 2)):
            step()
        calls = max(60, args.steps)
        fence()
        same_started = time.perf_counter()
        for _ in range(calls):
            step()
        fence()
        same_elapsed = time.perf_counter() - same_started
        profile = engine.last_call_profile()
        same_tapes = {"what": "the same tapes again and again: the plan of the previous call is re-used behind a guard, no planner kernel",
                      "ms_per_step": round(same_elapsed / calls * 1e3, 4), "value": round(profile.cells * calls / same_elapsed / 1e9, 1),
                      "unit": "GCUPS", "calls": calls, "planner_mode": int(profile.planner)}

    for index in range(args.warmup):
        steps_of[index % len(steps_of)]()
    cells_of_step = []
    for one in steps_of:  # cells per call of each batch (untimed; also leaves every plan shape warm)
        one()
        cells_of_step.append(float(engine.last_call_profile().cells))
    kernel_ms, planners, timed_cells = [], set(), 0.0
    fence()
    started = time.perf_counter()
    for index in range(args.steps):
        steps_of[index % len(steps_of)]()  # synchronous: returns after the scope's stream has drained
        kernel_ms.append(engine.last_call_profile().kernel_milliseconds)
        planners.add(int(engine.last_call_profile().planner))
        timed_cells += cells_of_step[index % len(steps_of)]
    fence()
    elapsed = time.perf_counter() - started
    if world > 1:
        slowest = torch.tensor([elapsed], dtype=torch.float64, device=where)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        elapsed = float(slowest)
    step()  # untimed: the headline batch's own matrix is what the checksum and the CPU baseline look at

    profile = engine.last_call_profile()

    cells_per_rank = torch.tensor([timed_cells], dtype=torch.float64, device=where)  # over the K timed steps
    checksum = results.sum().reshape(1).to(torch.float64)
    if world > 1:
        dist.all_reduce(cells_per_rank)
        dist.all_reduce(checksum)
    total_cells = float(cells_per_rank)

    if world > 1 and extras:
        fence()
        if rank == 0:  # single-process C driver over the same GPUs, while the other ranks wait
            for config in extras:
                try:
                    record = measure_c_node(config, [0] * world if args.same_device else list(range(world)), args)
                    if record is not None:
                        records.append(record)
                except Exception as problem:
                    records.append({"config": config, "entry_point": "szs_rocm_node_scores_u32tape", "error": repr(problem)})
        fence()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells / elapsed / 1e9
        kernel = float(np.mean(kernel_ms)) * 1e-3  # seconds per launch group, hipEvent pair on the library's stream
        gpu_matrix = results.cpu().numpy()
        valu_table, valu_where = _profile_json("valu_peak.json")
        pure = (valu_table or {}).get("myers_pure_W4_Tcells")
        lengths = load.queries.lengths().astype(np.int64)
        padded_cells = float((np.maximum(1, -(-lengths // 32)) * 32).sum()) * float(load.candidates.lengths().sum())
        line_roofline = roofline(args.config, profile, kernel, args.hbm_traffic_bytes)
        try:
            line_roofline["peak_measured"] = measured_hbm_peak(where)
            line_roofline["frac_of_measured_peak"] = round(line_roofline["achieved"] / line_roofline["peak_measured"], 6)
            line_roofline["peak_measured_how"] = "device-to-device copy of 1 GiB on this box, read + write bytes over the best of five (torch, HIP events)"
        except Exception as problem:
            line_roofline["peak_measured"] = None
            line_roofline["peak_measured_how"] = repr(problem)
        line_roofline["note"] = ("integer-VALU bound by construction (HBM traffic is a few % of the algorithmic bytes: tapes are "
                                 "L2-resident); the HBM fraction is reported because the metric asks for it, the `valu` and "
                                 "`myers_ceiling` objects are the rooflines that say something about the kernel")
        if load.kind == "levenshtein" and pure:
            line_roofline["myers_ceiling"] = {
                "bound": "integer VALU issue, measured", "achieved_Tcells_per_s_full_width": round(padded_cells / kernel / 1e12, 2),
                "peak_Tcells_per_s_full_width": pure, "frac": round(padded_cells / kernel / 1e12 / pure, 4),
                "peak_source": f"scripts/valu_peak.hip `myers_pure_W4_Tcells`: the kernel's column update on "
                               f"register-resident masks ({valu_where})",
                "useful_fraction_of_width": round(float(profile.cells) / padded_cells, 4)}
        line = {
            "metric": "DP cell-updates/s (GCUPS) on 1M-pair Levenshtein batch" if args.config == 2 else f"DP cell-updates/s (GCUPS), {load.name}",
            "value": round(value, 1), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {0: "u32 bit-vectors (u64 results)", 16: "i16 cells, two per VALU op (i64 results)", 32: "i32 cells (i64 results)",
                      64: "i64 cells"}.get(int(profile.cell_bits), "u32"),
            "data": "synthetic",
            "config": {"workload": load.name, "pairs_per_gpu": rows * columns, "cells_per_gpu": int(profile.cells),
                       "sharding": "query row blocks, candidates replicated" if world > 1 else "single GPU",
                       "entry_point": ENTRY_POINTS[load.kind], "generator": args.generator},
            "roofline": line_roofline,
            "host_overhead_ms_per_step": round(ms_per_step - kernel * 1e3, 4),
            "planner": " / ".join({0: "host", 1: "device", 2: "device, launches speculated on the previous call's shape",
                                   3: "plan of the previous call re-used for the same tapes, validated in the kernels"}[mode] for mode in sorted(planners)),
            "results_checksum": float(checksum),
            # the three figures the reference's own bench prints per engine (bench/similarities.cuh:344-366: bytes passed, operations =
            # cells, inputs processed, and the device-measured "Kernel" line :303-308)
            "reference_style": {"throughput_gb_s": round(float(load.queries.lengths().sum() * columns + load.candidates.lengths().sum() * rows) * world / (elapsed / args.steps) / 1e9, 1),
                                "efficiency_gops_s": round(value, 1), "pairs_per_second": round(rows * columns * world / (elapsed / args.steps), 0),
                                "kernel_gcups": round(float(profile.cells) / kernel / 1e9, 1), "kernel_ms": round(kernel * 1e3, 4)},
        }
        line["config"]["stream"] = ("two different batches of this shape alternate: every timed call plans its tapes afresh on the device"
                                    if len(steps_of) > 1 else "the same batch every step")
        if same_tapes is not None:
            line["config"]["same_tapes_gcups"] = same_tapes["value"]  # kept inside `config` so that a truncated record still has it
            line["same_tapes"] = same_tapes
        if not args.no_cpu_baseline:  # rank 0, whatever N: the host cores are this box's
            line["cpu_baseline"] = cpu_baseline(load, gpu_matrix, args.cpu_seconds)
            attach_cpu_baselines(records, args.cpu_seconds / 2)  # half the headline's budget each: the default run stays within ~2.5 minutes
        for record in records:
            record.pop("_cpu_baseline_inputs", None)
        if records:
            line["configs"] = records
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
