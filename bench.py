#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's configs, through the C-ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2] [--extra-configs 3,4,5,6] [--no-cpu-baseline]

`--gpus N` with N > 1 starts its own N ranks (one process per GPU under torch.distributed.run, 127.0.0.1, a free port); under a
launcher that already set WORLD_SIZE (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`) it
is one of those ranks.  Fewer than N GPUs visible: one JSON line with "error", exit code 2.

Metric : DP cell-updates/s (GCUPS) = sum over scored pairs of len(query) * len(candidate) / seconds / 1e9
         (the reference's accounting, /root/reference/bench/similarities.cuh:344-366).
Step   : one `szs_levenshtein_distances_u32tape` call - the whole synchronous C-ABI call, host planning included - over
         config 2: a 1024 x 1024 cross-product (1,048,576 pairs) of printable-ASCII strings, length U[96,160], unit costs.
         Tapes and the results matrix are resident in HBM before the timed region starts.
N > 1  : one process per GPU.  The HEADLINE shards by QUERY ROW BLOCKS: every rank scores its own
         1024 query rows against the same 1024 candidates (broadcast once over RCCL/xGMI before timing), so per-GPU work
         is fixed: "weak" scaling, and the timed path has no collective (rows are independent; results stay sharded).
Line   : the LAST stdout line of rank 0 is the headline alone, under 4 KB (tests/test_bench_line.py).  `roofline` prices the
         dominant kernel against HBM with ALGORITHMIC bytes (272 B per pair at len 128: len(q) + len(c) + 2 offsets + one
         8-byte result) over the hipEvent-measured kernel time the library records on its own stream.
         `cpu_baseline` times the reference's own SIMD engines (oracle/_ref, built from /root/reference) on this box's host
         cores - a reported baseline only.
configs: BEFORE the headline, one `{"configs_record": {...}}` line per other config (9: config 2 at a fixed 128 bytes, its
         "peak" variant; 3: NW BLOSUM62; 4: SW affine NUC.4.4; 5: byte-level Levenshtein on Zipf UTF-8; 6 = 5u: the same at the
         codepoint level; 7 / 8: config 2's batch under non-unit costs, linear 1/3/3 and affine 0/1/4/2; 10: 4096 x 4096 tiny
         tokens), each timed through its own C-ABI entry point with kernel and wall GCUPS, checksum, HBM roofline, the VALU
         counters of its dominant kernel (committed PMC passes) and the reference's Ice Lake engine as `cpu_baseline` on a
         stated sample whose cells are also compared with the GPU's; all of it also in gpurun_out/bench_configs.json.  The
         headline keeps their wall GCUPS as `configs_gcups`.  With N > 1 configs 4 and 5 are STRONG-scaled the way
         BASELINE.json specifies them: ONE batch, rows dealt over the ranks by LPT (`stringzilla_amd/sharded.py`), per-GPU busy
         time, imbalance = max / mean, aggregate GCUPS - plus the same batch through the single-process C entry
         `szs_rocm_node_*` (one host thread per GPU).
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
PROFILE_DIRS = [os.path.join(ROOT, "profiles", name) for name in ("r06", "r05", "r04", "r03", "r02", "r01")]
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


def _by_prefix(table, prefix):
    """The entry of a committed summary whose key is `prefix` or `prefix<...>` (kernel templates are keyed with their arguments:
    round 5 looked `cfg11:fingerprint_segments_kernel` up by its bare name, found nothing and printed an ESTIMATE at frac 1.16)."""
    for key in sorted(table or {}):
        if key == prefix or key.startswith(prefix + "<"):
            return key, table[key]
    return None, None


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--config", type=int, default=2, help="BASELINE.json config index of the headline (2 = the metric's config)")
    parser.add_argument("--generator", default="mt19937_64", choices=["numpy", "mt19937_64"],
                        help="where configs 1-4 come from: std::mt19937_64, the generator SURVEY.md section 8(d) names "
                             "(csrc/workloads/workloads_mt19937.cpp spells the mapping out; reproducible from C++), or numpy's "
                             "default_rng (the batches rounds 1-3 were profiled on: the same shapes, other strings).  Configs 5 / 5u "
                             "(Zipf UTF-8) exist in numpy only.  A run whose generator is not available FAILS; it never scores another batch "
                             "under the same config name")
    parser.add_argument("--extra-configs", default=None,
                        help="comma-separated configs reported in the `configs` array (default: 3,4,5,6,7,8 on one GPU, "
                             "4,5 strong-scaled on several; 'none' to skip)")
    parser.add_argument("--extra-seconds", type=float, default=2.0, help="GPU time budget per extra config")
    parser.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU time budget of the headline's cpu_baseline sample")
    parser.add_argument("--extra-cpu-seconds", type=float, default=2.5,
                        help="CPU time budget of each `configs` record's cpu_baseline (a config whose smallest fair sample - a row per "
                             "thread x one SIMD lane group - takes longer gets three runs of that sample)")
    parser.add_argument("--details", default=os.path.join("gpurun_out", "bench_configs.json"),
                        help="where the `configs` records and the untrimmed headline are written (also printed, one JSON line each, "
                             "BEFORE the headline; the LAST stdout line is the headline alone, under 4 KB)")
    parser.add_argument("--extra-scale", type=float, default=1.0,
                        help="testing aid: shrinks the matrix side of the `configs` records (1.0 = BASELINE.json's sizes)")
    parser.add_argument("--verify-seconds", type=float, default=8.0,
                        help="CPU time budget per config for checking the timed batch CELL FOR CELL against the reference's engines (the "
                             "reference's own bench does, bench/similarities.cuh:410-423): the whole matrix when the checker's measured rate "
                             "says it fits, else as many whole rows, evenly spaced over the queries, as do")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--fingerprints-only", action="store_true",
                        help="time `szs_fingerprints_u32tape` alone and print its record (what scripts/profile_configs.sh runs as config 11)")
    parser.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    parser.add_argument("--same-device", action="store_true",
                        help="testing aid: every rank uses cuda:0 (with --backend gloo), to exercise the N > 1 code "
                             "path on a one-GPU box; the numbers of such a run mean nothing")
    parser.add_argument("--force-distributed", action="store_true",
                        help="testing aid: run the N > 1 code path (torch.distributed over --backend, strong-scaled records, the C node driver) "
                             "also with --gpus 1: RCCL's first contact on a one-GPU box")
    parser.add_argument("--hbm-traffic-bytes", type=float, default=None,
                        help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: the committed summary "
                             "profiles/rNN/pmc_summary.json of this same command (scripts/profile_gpu.sh)")
    parser.add_argument("--c-node-leg", default=None, metavar="CONFIG:DEVICE,DEVICE,...",
                        help="internal: ONE strong-scaled record through the single-process C node driver, printed as a JSON line (the N > 1 "
                             "run calls itself with this in a child process: first contact with real peers must not cost the headline line)")
    return parser.parse_args()


def resolve_generator(wanted):
    """`mt19937_64` needs stringzilla_amd/lib/libszs_workloads_mt19937.so (csrc/Makefile builds it beside the scoring library).
    Without it the run FAILS: numpy batches under the same config name would be a different measurement."""
    from stringzilla_amd import workloads

    if wanted == "mt19937_64" and not os.path.exists(workloads.MT19937_64_LIBRARY):
        raise SystemExit(json.dumps({"error": f"--generator mt19937_64 needs {os.path.relpath(workloads.MT19937_64_LIBRARY, ROOT)} "
                                              f"(make -C stringzilla_amd/csrc); pass --generator numpy to score numpy batches"}))
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

def cpu_baseline(load, gpu_matrix, seconds, cell_bits=0, with_serial=True, verify_seconds=0.0):
    """Times the reference's own CPU engines (best SIMD tier, all host threads) on a BOUNDED sample of the same batch -
    evenly spaced query rows x evenly spaced candidates, sized from its first (verified) run to about `seconds` of CPU work,
    the whole batch when that fits - and checks that they produce the very cells the GPU produced.  Test-infrastructure
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

    def run(rows, columns, engine=None):
        engine = engine or checker
        queries = [load.queries[int(i)] for i in rows]
        candidates = [load.candidates[int(j)] for j in columns]
        if load.kind == "levenshtein":
            return engine.levenshtein(queries, candidates, **load.costs)
        if load.kind == "levenshtein_utf8":
            return engine.levenshtein_utf8(queries, candidates, **load.costs)
        scorer = engine.needleman_wunsch if load.kind == "needleman_wunsch" else engine.smith_waterman
        return scorer(queries, candidates, *matrices.by_name(load.table), **load.costs)

    def verified(rows, columns, engine=None):
        started = time.perf_counter()
        matrix = run(rows, columns, engine)
        elapsed = time.perf_counter() - started
        assert np.array_equal(matrix.view(np.int64), gpu_matrix[np.ix_(rows, columns)].view(np.int64)), "CPU baseline and GPU disagree"
        return elapsed

    q_lengths, c_lengths = load.queries.lengths(), load.candidates.lengths()
    spaced = lambda count, take: np.unique(np.linspace(0, count - 1, num=max(1, min(count, take))).round().astype(np.int64))
    cells_of = lambda rows, columns: float(q_lengths[rows].sum()) * float(c_lengths[columns].sum())

    # The smallest fair sample: one row per thread (the shim deals contiguous row blocks to threads) x one SIMD lane group of
    # candidates - the reference's engines score one query against 16 (32-bit cells) / 32 / 64 candidates at once, one per lane,
    # so a sample's candidates come in whole lane groups (round 2's config-4 sample of 26 candidates left a third of the lanes
    # empty and read 26 GCUPS for ~32).  Its first run is verified against the GPU's cells and sizes everything after it.
    lanes = 16 if int(cell_bits) == 32 else 64
    rows, columns = spaced(len(q_lengths), max(cores, 1)), spaced(len(c_lengths), lanes)
    first = verified(rows, columns)
    if first < seconds / 5:  # grow towards a fifth of the budget per run: first more candidates (whole lane groups), then more rows
        budget_cells = cells_of(rows, columns) / max(first, 1e-4) * seconds / 5
        per_column = cells_of(rows, np.arange(len(c_lengths))) / len(c_lengths)
        take_columns = int(min(len(c_lengths), max(lanes, budget_cells / max(per_column, 1.0) // lanes * lanes)))
        columns = spaced(len(c_lengths), take_columns)
        if take_columns == len(c_lengths):
            per_row = cells_of(np.arange(len(q_lengths)), columns) / len(q_lengths)
            rows = spaced(len(q_lengths), int(min(len(q_lengths), max(len(rows), budget_cells / max(per_row, 1.0)))))
        first = verified(rows, columns)
    whole = len(rows) == len(q_lengths) and len(columns) == len(c_lengths)
    # A run lasts a quarter of a second or more: a batch the CPU finishes in 30 ms (configs 2, 7, 8) is passed several times back
    # to back per run - timed one pass at a time, 256 threads starting up made the median wander between 340 and 580.  Five or
    # more runs when the budget allows; a batch whose smallest fair sample already takes a third of the budget gets three, the whole budget: two.
    passes = 1 if first >= 0.25 else int(min(64, np.ceil(0.25 / max(first, 1e-4))))
    repeats = int(max(3 if first * passes > seconds / 5 else 5, min(20, (seconds - first) / max(first * passes, 1e-3))))
    if first > seconds:  # one run of the smallest fair sample is already the whole budget (config 4: ~10 s): the verified run and one more
        repeats = 2
    runs = [first] if passes == 1 and first * 3 > seconds else []  # ... the verified run counting as the first of them
    while len(runs) < repeats:
        started = time.perf_counter()
        for _ in range(passes):
            run(rows, columns)
        runs.append((time.perf_counter() - started) / passes)
    elapsed = float(np.median(runs))  # the median: a 256-thread host shows stragglers, a mean of two runs wandered by 25 %
    record = {
        "value": round(cells_of(rows, columns) / elapsed / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": kind, "tier": label,
        "spread": [round(cells_of(rows, columns) / max(runs) / 1e9, 2), round(cells_of(rows, columns) / min(runs) / 1e9, 2)],
        "sample": (f"full {len(q_lengths)}x{len(c_lengths)} batch" if whole else f"{len(rows)} spaced rows x {len(columns)} spaced candidates")
                  + f"; median of {len(runs)} runs" + (f" x {passes} passes" if passes > 1 else "") + "; cells verified equal to the GPU's",
        "sample_rows": len(rows), "sample_columns": len(columns), "runs": len(runs), "passes_per_run": passes, "verified": True,
        # how much of the host the sample keeps busy: the shim deals contiguous row blocks to threads, a thread scores its rows against
        # lane groups of 16 / 64 candidates (a sample of one row per thread x one lane group reads far below a whole batch: say so)
        "threads_busy": int(min(cores, len(rows))), "rows_per_thread": round(len(rows) / max(1, min(cores, len(rows))), 2),
        "lane_groups_per_row": round(len(columns) / lanes, 2),
    }
    # ---- the timed batch CELL FOR CELL (bench/similarities.cuh:410-423 checks every batch it times), once and untimed: the whole
    # matrix when the checker's measured rate says it fits `verify_seconds`, else as many WHOLE rows - evenly spaced over the
    # queries, never fewer than 64 - as do.  The candidates take the row role in the checker (its threads share rows; 64 rows would
    # keep 64 of 256 threads busy) - every table and cost scheme of these configs is symmetric, so the matrix is the transpose.
    if whole:
        record["checked"] = {"whole_matrix": True, "rows": len(q_lengths), "columns": len(c_lengths)}
    elif verify_seconds > 0:
        everything, every_column = np.arange(len(q_lengths)), np.arange(len(c_lengths))
        rate = cells_of(rows, columns) / max(elapsed, 1e-6)
        per_row = cells_of(everything, every_column) / max(len(q_lengths), 1)
        take = int(min(len(q_lengths), max(min(64, len(q_lengths)), rate * verify_seconds / max(per_row, 1.0))))
        check_rows = everything if take >= len(q_lengths) else spaced(len(q_lengths), take)
        started = time.perf_counter()
        candidates_as_rows = [load.candidates[int(j)] for j in every_column]
        picked = [load.queries[int(i)] for i in check_rows]
        if load.kind == "levenshtein":
            transposed = checker.levenshtein(candidates_as_rows, picked, **load.costs)
        elif load.kind == "levenshtein_utf8":
            transposed = checker.levenshtein_utf8(candidates_as_rows, picked, **load.costs)
        else:
            scorer = checker.needleman_wunsch if load.kind == "needleman_wunsch" else checker.smith_waterman
            transposed = scorer(candidates_as_rows, picked, *matrices.by_name(load.table), **load.costs)
        wrong = int((transposed.T.view(np.int64) != gpu_matrix[check_rows].view(np.int64)).sum())
        assert wrong == 0, f"{wrong} cells of {load.name} differ from the reference's"
        record["checked"] = {"whole_matrix": len(check_rows) == len(q_lengths), "rows": len(check_rows), "columns": len(c_lengths),
                             "seconds": round(time.perf_counter() - started, 2)}
    # the reference reports its Serial engine beside the SIMD tiers (similarities/README.md:21-24): ONE thread of the serial tier
    # on a sample of about a second (four rows, one lane group of candidates)
    if with_serial and kind == "reference":
        try:
            plain = binding.reference(tier=0, threads=1)
            serial_rows, serial_columns = spaced(len(q_lengths), 4), spaced(len(c_lengths), lanes)
            once = verified(serial_rows, serial_columns, plain)
            again = int(max(1, min(50, 1.0 / max(once, 1e-4))))
            started = time.perf_counter()
            for _ in range(again):
                run(serial_rows, serial_columns, plain)
            record["serial_1_thread"] = {"value": round(cells_of(serial_rows, serial_columns) * again / (time.perf_counter() - started) / 1e9, 3),
                                         "unit": "GCUPS", "cores": 1, "sample": f"{len(serial_rows)} rows x {len(serial_columns)} candidates x {again} passes, serial tier, verified"}
        except AssertionError:
            raise
        except Exception as problem:
            record["serial_1_thread"] = {"error": repr(problem)}
    return record


# ---- rooflines -------------------------------------------------------------------------------------------------------------

def roofline(config, profile, kernel_seconds, traffic_override=None, leg=None):
    """HBM roofline from the ALGORITHMIC bytes of one call over the kernel time measured live (hipEvent pair on the library's
    stream), beside what the committed rocprofv3 --pmc passes of this same command saw: HBM bytes actually moved per call
    (`traffic`: (FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernels of one call, raw) and - what actually binds this path -
    VALU issue and LDS occupancy (scripts/profile_configs.sh -> profiles/rNN).  Keys only, no prose: the line must stay short
    (DESIGN.md section 5 says what each key means)."""
    achieved = profile.algorithmic_bytes / kernel_seconds / 1e9
    record = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic_override, "kernel_ms": round(kernel_seconds * 1e3, 4),
              "algorithmic_bytes": int(profile.algorithmic_bytes), "launches_per_step": int(profile.launches),
              "kernel_gcups": round(profile.cells / kernel_seconds / 1e9, 1)}
    summary, where = _profile_json("pmc_configs.json")  # "cfgN:kernel" and "cfgN:__call__" (scripts/pmc_configs.py)
    # `leg`: a config whose calls are of two kinds (config 2: "fresh" batches run the launch that plans itself, "same_tapes" the plain
    # one behind the guard) has one record per kind beside the blend of the run the counters were taken on (scripts/pmc_configs.py);
    # the counters joined to a timed leg are those of the kernel THAT leg launches
    call = (summary or {}).get(f"cfg{config}:__call__@{leg}") if leg else None
    record["pmc_leg"] = leg if call else None
    call = call or (summary or {}).get(f"cfg{config}:__call__")
    counted_on = (summary or {}).get("_library_sha256")
    # pmc_stale True: the counters below were taken on another build of the library than the one this run executes
    record["pmc_stale"] = None if not counted_on else counted_on != library_digest()
    if not call:
        record["traffic_source"] = "none: no committed PMC pass for this config" if traffic_override is None else "--hbm-traffic-bytes"
        return record
    record["pmc_source"] = where
    kernels = call.get("kernels", {})
    if kernels:
        dominant = max(kernels, key=lambda name: kernels[name]["share_of_kernel_time"])
        record["kernel"] = dominant
        if len(kernels) > 1:
            record["kernels_in_call"] = len(kernels)
    if traffic_override is None and "hbm_fetch_bytes_raw" in call and "hbm_write_bytes_raw" in call:
        record["traffic"] = round(call["hbm_fetch_bytes_raw"] + call["hbm_write_bytes_raw"])
    if "SQ_INSTS_VALU" in call:
        lane_ops = call["SQ_INSTS_VALU"] * 64.0
        mixes, _ = _profile_json("opcode_mix.json")
        # the ceiling of the call = its kernels' ceilings weighted by their share of the kernel time:
        # (F + H) / (F / 78.6 T + H / 39.3 T), F / H = full- / half-rate instructions of a kernel's main loop (scripts/opcode_mix.py)
        weights = {name: kernels[name]["share_of_kernel_time"] for name in kernels if (mixes or {}).get(name)}
        if weights:
            total = sum(weights.values())
            ceiling = 1e12 * total / sum(share / mixes[name]["ceiling_Tlane_ops_per_s"] for name, share in weights.items())
            main = max(weights, key=weights.get)
            full, half = mixes[main]["full_rate"], mixes[main]["half_rate"]
        else:
            ceiling, full, half = VALU_HALF_RATE_PEAK, None, None
        fraction = lambda key: round(call[key], 4) if key in call else None
        record["valu"] = {
            "bound": "int VALU issue, class-weighted", "frac": round(lane_ops / kernel_seconds / ceiling, 4),
            "achieved_Tlane_ops_per_s": round(lane_ops / kernel_seconds / 1e12, 2), "peak_Tlane_ops_per_s": round(ceiling / 1e12, 2),
            "lane_ops_per_cell": round(lane_ops / max(float(profile.cells), 1.0), 4),
            "wave_instructions_per_call": round(call["SQ_INSTS_VALU"]), "full_rate": full, "half_rate": half,
            "lds_busy": fraction("lds_busy_fraction"), "lds_conflict": fraction("lds_conflict_fraction"),
            "wait_to_issue": fraction("wave_wait_inst_fraction"), "parked": fraction("wave_wait_any_fraction"),
        }
    return record


def words_roofline(counted, bytes_moved, kernel_seconds, launches, planner_mode, peaks):
    """Config 10's record: the HBM-shaped figures (`bytes_moved` = the results matrix + both tapes + offsets, once) over the kernel
    time, beside what the committed PMC passes of this config saw (`counted` = roofline()'s record: HBM bytes per call, VALU issue,
    staleness) and - `issue_floor` - the launch's own wavefront-instructions at the pace its column update issues at when nothing
    else is in the way (`peaks` = profiles/rNN/valu_peak.json, scripts/valu_peak.hip: 192 instructions a wavefront-column of 2048
    pair-columns, 1024 SIMDs)."""
    record = {"bound": "hbm", "achieved": round(bytes_moved / kernel_seconds / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": round(bytes_moved / kernel_seconds / 1e9 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": int(bytes_moved),
              "algorithmic_bytes_are": "the results matrix + both tapes + offsets, once", "kernel_ms": round(kernel_seconds * 1e3, 4),
              "launches_per_step": launches, "traffic": counted.get("traffic"), "pmc_stale": counted.get("pmc_stale"),
              "pmc_source": counted.get("pmc_source"), "kernel": counted.get("kernel", "levenshtein_tiny_kernel"), "planner_mode": planner_mode}
    valu = counted.get("valu")
    if valu:
        record["valu"] = valu
    pace = (peaks or {}).get("tiny_pure_R16_Tpair_columns")
    if valu and pace and valu.get("wave_instructions_per_call"):
        floor_seconds = valu["wave_instructions_per_call"] * 2048.0 / (pace * 1e12) / 192.0
        record["issue_floor"] = {"bound": "its own instructions at the measured pace of its column (valu_peak.hip)",
                                 "floor_ms": round(floor_seconds * 1e3, 4), "frac": round(floor_seconds / kernel_seconds, 4)}
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
    if config in (10, 12):
        # Tiny tokens (12: at the codepoint level, one pass ahead of the launch writes the strings as bytes of rune ids): the one regime of this path where HBM is the roofline that binds - the RESULT MATRIX (8 bytes a pair against
        # ~13 bytes of strings per pair-row).  Algorithmic bytes here = results + both tapes + their offsets, once each.
        bytes_moved = len(queries) * len(candidates) * 8 + int(load.queries.lengths().sum() + load.candidates.lengths().sum()) + 4 * (len(queries) + len(candidates) + 2)
        record["roofline"] = words_roofline(record["roofline"], bytes_moved, kernel, int(profile.launches), int(profile.planner), _profile_json("valu_peak.json")[0])
    if config == 4:
        # Config 4's whole matrix is ~150 s of the reference's engines on this host (64 whole rows are checked against them below, in
        # `cpu_baseline.checked`).  ALL 262,144 cells are checked here against an independent implementation on the device: the
        # one-pair-per-lane kernel (`team` knob 0: 32-bit cells, another recurrence layout, hip/weighted.hip), untimed.
        from stringzilla_amd import _abi

        other = torch.empty_like(results)
        previous = _abi.tuning_set("team", 0)
        try:
            make_step(engine, scope, load, queries, candidates, other, device_index)()
            other_tier = engine.last_call_profile()
            record["cross_tier_check"] = {"against": "one pair per lane, 32-bit cells (team knob 0)", "cell_bits": int(other_tier.cell_bits),
                                          "team": int(other_tier.team), "whole_matrix": True, "cells_equal": bool(torch.equal(other, results))}
        finally:
            _abi.tuning_set("team", previous)
        assert record["cross_tier_check"]["cells_equal"], "config 4: the team tier and the one-pair-per-lane kernel disagree"
        make_step(engine, scope, load, queries, candidates, results, device_index)()  # (the engine's remembered plan is the team tier's again)
        del other
    if with_cpu:  # timed later, after every GPU measurement of the run (the host cores are busy for ~10 s per baseline); the
        # matrix stays in HBM until then - downloading 80 MB here would idle the shader engines right before the headline
        record["_cpu_baseline_inputs"] = (load, results)
    return record


def fingerprints_roofline(text_bytes, dimensions, documents, wall, summary, where, mixes, digest):
    """The VALU pricing of `szs_fingerprints_u32tape`'s record from the committed passes (`summary` = pmc_configs.json, `mixes` =
    opcode_mix.json; keys looked up by PREFIX - the kernel is a template): instructions COUNTED by the PMC pass of this very call
    when there is one, else estimated from the assembly's main loop (four positions an iteration) and labelled so."""
    _, counters = _by_prefix(summary, "cfg11:fingerprint_segments_kernel")
    counted = (counters or {}).get("SQ_INSTS_VALU")
    mix_key, mix = _by_prefix(mixes, "fingerprint_segments_kernel")
    ceiling = mix["ceiling_Tlane_ops_per_s"] * 1e12 if mix else VALU_HALF_RATE_PEAK
    if counted:
        lane_ops, how = counted * 64.0, f"PMC: SQ_INSTS_VALU of the segments kernel, {where}"
    else:
        per_position = mix["valu_instructions"] / 4.0 if mix else 25.0
        lane_ops, how = per_position * text_bytes * dimensions, "ESTIMATE: main loop of the assembly / 4 positions" if mix else "ESTIMATE: 25 assumed"
    counted_on = (summary or {}).get("_library_sha256")
    moved = text_bytes + 8 * dimensions * documents
    return {"bound": "int / fp64 VALU issue", "counted": how, "kernel": mix_key or "fingerprint_segments_kernel",
            "lane_ops_per_byte_and_dimension": round(lane_ops / (text_bytes * dimensions), 2),
            "pmc_stale": None if not (counted and counted_on) else counted_on != digest,
            "achieved_Tlane_ops_per_s": round(lane_ops / wall / 1e12, 2), "peak_Tlane_ops_per_s": round(ceiling / 1e12, 2),
            "frac": round(lane_ops / wall / ceiling, 4),
            "hbm": {"algorithmic_bytes": moved, "achieved_gb_s": round(moved / wall / 1e9, 2), "peak": HBM_PEAK_GBPS}}


def fingerprints_cpu_baseline(texts, dimensions, gpu_hashes, gpu_counts, seconds):
    """The reference's own SIMD hashers (`floating_rolling_hashers<sz_cap_skylake_k / haswell / serial, 64>` through oracle/_ref,
    documents dealt over every host thread) on a BOUNDED sample of the same documents, their sketches compared with the GPU's."""
    from oracle import binding

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    tier = binding.reference_best_tier()
    documents = [texts[i] for i in range(len(texts))]

    def run(count):
        started = time.perf_counter()
        hashes, counts, ran = binding.reference_fingerprints_tiered(documents[:count], dimensions, tier, cores)
        elapsed = time.perf_counter() - started
        assert np.array_equal(hashes, gpu_hashes[:count]) and np.array_equal(counts, gpu_counts[:count]), "CPU and GPU fingerprints differ"
        return elapsed, ran

    count = min(len(documents), max(cores, 1))  # a document per thread first; then as many as the budget holds
    first, ran = run(count)
    count = int(min(len(documents), max(count, count * seconds / 3 / max(first, 1e-4))))
    runs = [run(count)[0] for _ in range(3)]
    elapsed = float(np.median(runs))
    sample_bytes = sum(len(document) for document in documents[:count])
    return {"value": round(sample_bytes * dimensions / elapsed / 1e12, 4), "unit": "10^12 byte-dimensions/s", "cores": cores, "kind": "reference",
            "tier": {0: "serial", 1: "haswell (AVX2)", 2: "skylake (AVX-512)"}[ran], "threads_busy": int(min(cores, count)),
            "sample": f"the first {count} of {len(documents)} documents, {dimensions} dimensions; median of {len(runs)} runs; sketches verified equal to the GPU's",
            "verified": True}


def measure_fingerprints(scope, device_index, args, fence):
    """`szs_fingerprints_u32tape`: rolling MinHash over 1024 documents of ~10 KB, 1024 dimensions of the
    reference's default window widths - bytes of text per second and byte-dimensions per second of the whole C-ABI call.  The
    kernel's work is 25 instructions per byte and dimension: VALU-bound, the text is read once per 256
    dimensions from LDS."""
    import torch

    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    from stringzilla_amd import _abi

    dimensions = 1024
    texts = workloads.random_tape(np.random.default_rng(11), 1024, 8192, 12288, workloads.ASCII_PRINTABLE).to_device(device_index)
    engine = szs.Fingerprints(dimensions, capabilities=scope)
    # the raw C-ABI call into matrices that live in HBM, like every other record (round 4 timed the Python wrapper, which also
    # downloads 8 MB into fresh pageable NumPy arrays per call: its "first pass" ran at twice the time of every later one -
    # host pages being faulted in, not the kernel, whose dispatches take 5.8 ms from the sixth on and 6.5 the very first:
    # profiles/r05/fingerprints_first_pass.txt)
    outputs = torch.empty((2, len(texts), dimensions), dtype=torch.int32, device=torch.device("cuda", device_index))
    tape, error = texts._tape(device_index), ctypes.c_char_p()

    def step():
        status = _abi.lib.szs_fingerprints_u32tape(engine.handle, scope.handle, ctypes.byref(tape), outputs[0].data_ptr(), dimensions * 4,
                                                   outputs[1].data_ptr(), dimensions * 4, ctypes.byref(error))
        if status:
            raise RuntimeError(f"szs_fingerprints_u32tape failed: {status} {error.value}")

    step()
    fence()
    started = time.perf_counter()
    step()
    once = time.perf_counter() - started
    repeats = int(max(3, min(50, args.extra_seconds / max(once, 1e-4))))
    walls = []
    for _ in range(2):  # two passes; both are reported, the faster one counts
        fence()
        started = time.perf_counter()
        for _ in range(repeats):
            step()
        fence()
        walls.append((time.perf_counter() - started) / repeats)
    wall = min(walls)
    text_bytes = int(texts.lengths().sum())
    summary, where = _profile_json("pmc_configs.json")
    mixes, _ = _profile_json("opcode_mix.json")
    record = {"config": "fingerprints", "workload": "1024 ASCII documents of 8-12 KB, 1024 dimensions, default window widths",
              "entry_point": "szs_fingerprints_u32tape", "n_gpus": 1, "steps": repeats, "ms_per_step": round(wall * 1e3, 3),
              "passes_ms": [round(w * 1e3, 3) for w in walls],
              "value": round(text_bytes * dimensions / wall / 1e12, 3), "unit": "10^12 byte-dimensions/s",
              "text_gb_s": round(text_bytes / wall / 1e9, 2), "results_checksum": int(outputs[0].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item() % (1 << 53)),
              "roofline": fingerprints_roofline(text_bytes, dimensions, len(texts), wall, summary, where, mixes, library_digest())}
    if not args.no_cpu_baseline:  # timed later, with the other CPU baselines, after every GPU measurement of the run
        record["_fingerprints_inputs"] = (texts, dimensions, outputs)
    return record


def attach_cpu_baselines(records, seconds, verify_seconds=0.0):
    for record in records:
        sketched = record.pop("_fingerprints_inputs", None)
        if sketched is not None:
            try:
                sketches = sketched[2].cpu().numpy().view(np.uint32)
                record["cpu_baseline"] = fingerprints_cpu_baseline(sketched[0], sketched[1], sketches[0], sketches[1], seconds)
            except AssertionError:
                raise
            except Exception as problem:  # the checker is optional equipment; the GPU numbers stand without it
                record["cpu_baseline"] = {"error": repr(problem)}
        inputs = record.pop("_cpu_baseline_inputs", None)
        if inputs is None:
            continue
        try:
            record["cpu_baseline"] = cpu_baseline(inputs[0], inputs[1].cpu().numpy(), seconds, cell_bits=record.get("cell_bits", 0), with_serial=False,
                                                  verify_seconds=verify_seconds)
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
                         float(local.sum()) if len(rows) else 0.0, state.get("kernel", 0.0),
                         float(profile.algorithmic_bytes) if busy else 0.0], dtype=torch.float64,
                        device=where if dist.get_backend() == "nccl" else "cpu")
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank:
        return None
    seconds = np.array([float(g[0]) for g in gathered])
    cells = sum(float(g[1]) for g in gathered)
    slowest_kernel = max(float(g[4]) for g in gathered)
    moved = sum(float(g[5]) for g in gathered)  # algorithmic bytes of the whole batch: every rank's rows against every candidate
    return {
        # the job's HBM roofline at N GPUs: the batch's algorithmic bytes over the SLOWEST rank's kernel time against N x 8 TB/s
        "roofline": {"bound": "hbm", "achieved": round(moved / max(slowest_kernel, 1e-12) / 1e9, 2), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                     "frac": round(moved / max(slowest_kernel, 1e-12) / 1e9 / (HBM_PEAK_GBPS * world), 6), "algorithmic_bytes": int(moved),
                     "kernel_ms_slowest_rank": round(slowest_kernel * 1e3, 4)},
        "kernel_gcups": round(cells / max(slowest_kernel, 1e-12) / 1e9, 1), "backend": dist.get_backend(), "ranks": dist.get_world_size(),
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


def free_port():
    import socket

    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        return probe.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU under
    `torch.distributed.run`, rendezvous on 127.0.0.1 and a free port, the same argv) and become them.  A box with fewer than N
    devices gets a parseable line and a non-zero exit, not a traceback ("missing" #2)."""
    import torch

    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < (1 if args.same_device else args.gpus):
        print(json.dumps({"error": f"--gpus {args.gpus} but {visible} GPU(s) visible", "n_gpus": args.gpus, "visible_gpus": visible}), flush=True)
        raise SystemExit(2)
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, command)


def c_node_leg_in_a_child(config, devices, args):
    """measure_c_node() in a process of its own: one process driving every GPU of the node through peer copies is the part of the
    N > 1 run that no one-GPU box has exercised on distinct devices - whatever happens to it there (a fault, a hang), rank 0 still
    prints the headline, and the record says what happened."""
    import subprocess

    command = [sys.executable, os.path.abspath(__file__), "--c-node-leg", f"{config}:{','.join(str(d) for d in devices)}",
               "--extra-seconds", str(args.extra_seconds), "--extra-scale", str(args.extra_scale), "--generator", args.generator]
    environment = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
    try:
        done = subprocess.run(command, env=environment, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {"config": config, "entry_point": "szs_rocm_node_scores_u32tape", "error": "the child process did not finish in 300 s"}
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    if done.returncode != 0 or not lines:
        return {"config": config, "entry_point": "szs_rocm_node_scores_u32tape",
                "error": f"the child process ended with {done.returncode}: {done.stderr.strip()[-300:]}"}
    return json.loads(lines[-1])


def main():
    args = parse_args()
    if args.c_node_leg:  # (the child of c_node_leg_in_a_child)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
        config, devices = args.c_node_leg.split(":")
        args.generator = resolve_generator(args.generator)
        record = measure_c_node(int(config), [int(d) for d in devices.split(",")], args)
        print(json.dumps(record if record is not None else {"config": int(config), "entry_point": "szs_rocm_node_scores_u32tape", "error": "this build has no node driver"}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)  # does not return
    clock = {"started": time.perf_counter()}
    # The application's choice, made before HIP initialises: twelve hardware queues, so that the per-width launches of a
    # mixed-length batch (configs 5 / 5u: up to eight streams) each get their own.  The library itself never writes the
    # environment; it reads this variable when it is loaded and sizes its fan-out to it (csrc/host/tuning.c).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    import torch
    import torch.distributed as dist

    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    if args.force_distributed and args.gpus == 1 and "WORLD_SIZE" not in os.environ:  # a world of one rank, no launcher around it
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # `--force-distributed`: the N > 1 code path - process group, broadcast tapes, strong-scaled records through the sharded driver, the C
    # node driver, the barriers and all-reduces of the headline - with a world of ONE rank: what a one-GPU box can run of `--backend nccl`
    distributed = world > 1 or args.force_distributed
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", "n_gpus": args.gpus}), flush=True)
        raise SystemExit(2)
    if distributed:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
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
    if distributed:  # the one exchange step of the path: replicate the candidates tape over RCCL / xGMI, before timing;
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
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if args.fingerprints_only:
        alone = [measure_fingerprints(scope, local_rank, args, fence)]
        attach_cpu_baselines(alone, args.extra_cpu_seconds)
        print(json.dumps(alone[0]), flush=True)
        return
    # ---- the other configs run BEFORE the headline's warm-up: a run of W = 5 short warm-up steps on a GPU that has idled
    # through the set-up is timed on its clock ramp (round 1: 64.4 TCUPS with 20 steps against 67.9 with 200); after seconds
    # of NW / SW scoring the clocks are where a busy GPU keeps them.  The headline itself is unchanged: W untimed steps, then
    # exactly K timed ones.  (The single-process C driver of N > 1 runs last, while the other ranks wait.)
    if args.extra_configs is None:
        extras = [9, 3, 4, 5, 6, 7, 8, 10, 12] if not distributed else [2, 4, 5]  # N > 1: the metric's own batch and the two 8-GPU configs, STRONG-scaled
    else:
        extras = [] if args.extra_configs.strip().lower() in ("", "none") else [int(x) for x in args.extra_configs.split(",")]
    records = []
    for config in extras:
        try:
            if not distributed:
                records.append(measure_extra(config, scope, local_rank, args, fence, not args.no_cpu_baseline))
            else:  # configs 4 and 5 strong-scaled over the ranks; every rank takes part
                record = measure_strong(config, scope, local_rank, args, fence, dist, world, rank, where)
                if record is not None:
                    records.append(record)
        except AssertionError:
            raise
        except Exception as problem:  # an extra record must never cost the headline line
            records.append({"config": config, "error": repr(problem)})

    if not distributed and args.extra_configs is None:
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
        # (at least 400 calls, ~76 ms: the same-tapes figure is an average over them, and the power management has then settled on this
        # kernel's instruction mix before the headline's W + K steps begin - with 60 calls a run of K = 20 steps was timed on the
        # ramp: 87.2 TCUPS where K = 200 reads 89.9, the kernel itself 94.4 against 96.3; profiles/r06/bench_20_steps.jsonl)
        calls = max(400, args.steps)
        fence()
        same_started = time.perf_counter()
        for _ in range(calls):
            step()
        fence()
        same_elapsed = time.perf_counter() - same_started
        profile = engine.last_call_profile()
        # the same tapes again and again: the plan of the previous call is re-used behind a guard, no planner kernel
        same_tapes = {"ms_per_step": round(same_elapsed / calls * 1e3, 4), "value": round(profile.cells * calls / same_elapsed / 1e9, 1),
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
    if distributed:
        slowest = torch.tensor([elapsed], dtype=torch.float64, device=where)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        elapsed = float(slowest)
    # ---- every batch of the timed stream is CHECKED, outside the timed region (bench/similarities.cuh:410-423 checks every batch it
    # times): (a) the stream is replayed - the same calls in the same order - with the matrix summed on the device after every call
    # and compared with that batch's own sum; (b) below, on rank 0, each batch's whole matrix is compared cell for cell with the
    # reference's engines (`cpu_baseline.checked`, `stream_check`)
    sums_of_batches, matrices_of_batches = [], []
    for one in steps_of:
        one()
        sums_of_batches.append(int(results.sum().item()))
        if rank == 0 and not args.no_cpu_baseline and len(steps_of) > 1:
            matrices_of_batches.append(results.cpu().numpy().copy())
    replayed, replay_mismatches = min(args.steps, 200), 0
    for index in range(replayed):
        steps_of[index % len(steps_of)]()
        replay_mismatches += int(results.sum().item()) != sums_of_batches[index % len(steps_of)]
    assert replay_mismatches == 0, f"{replay_mismatches} of {replayed} replayed calls of the timed stream wrote another matrix"
    step()  # untimed: the headline batch's own matrix is what the checksum and the CPU baseline look at

    profile = engine.last_call_profile()

    device_ids = torch.zeros(world, dtype=torch.float64, device=where)
    device_ids[rank] = float(torch.cuda.current_device() + 1)
    if distributed:
        dist.all_reduce(device_ids)
    devices_in_use = len({int(v) for v in device_ids.tolist()})
    cells_per_rank = torch.tensor([timed_cells], dtype=torch.float64, device=where)  # over the K timed steps
    checksum = results.sum().reshape(1).to(torch.float64)
    if distributed:
        dist.all_reduce(cells_per_rank)
        dist.all_reduce(checksum)
    total_cells = float(cells_per_rank)

    if distributed and extras:
        fence()
        if rank == 0:  # single-process C driver over the same GPUs, while the other ranks wait
            for config in extras:
                try:
                    record = c_node_leg_in_a_child(config, [0] * world if args.same_device else list(range(world)), args)
                    if record is not None:
                        records.append(record)
                except Exception as problem:
                    records.append({"config": config, "entry_point": "szs_rocm_node_scores_u32tape", "error": repr(problem)})
        fence()

    clock["gpu_done"] = time.perf_counter()
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells / elapsed / 1e9
        kernel = float(np.mean(kernel_ms)) * 1e-3  # seconds per launch group, hipEvent pair on the library's stream
        gpu_matrix = results.cpu().numpy()
        valu_table, _ = _profile_json("valu_peak.json")
        pure = (valu_table or {}).get("myers_pure_W4_Tcells")
        lengths = load.queries.lengths().astype(np.int64)
        padded_cells = float((np.maximum(1, -(-lengths // 32)) * 32).sum()) * float(load.candidates.lengths().sum())
        # the counters joined to the timed leg are those of the kernel THAT leg launches: fresh batches plan themselves inside the
        # scoring launch (planner mode 4), the same tapes again run the plain launch behind the guard (mode 3)
        leg = "fresh" if planners == {4} else "same_tapes" if planners == {3} else None
        line_roofline = roofline(args.config, profile, kernel, args.hbm_traffic_bytes, leg=leg)
        try:  # device-to-device copy of 1 GiB on this box, read + write bytes over the best of five (torch, HIP events)
            line_roofline["peak_measured"] = measured_hbm_peak(where)
            line_roofline["frac_of_measured_peak"] = round(line_roofline["achieved"] / line_roofline["peak_measured"], 6)
        except Exception as problem:
            line_roofline["peak_measured"] = None
            line_roofline["peak_measured_error"] = repr(problem)[:80]
        if load.kind == "levenshtein" and pure:
            # the ceiling of the kernel's own column update on register-resident masks, measured: scripts/valu_peak.hip
            # `myers_pure_W4_Tcells`; cells counted at full bit-vector width (phantom rows included) on both sides
            line_roofline["myers_ceiling"] = {
                "bound": "int VALU issue, measured (valu_peak.hip)", "achieved_Tcells_per_s_full_width": round(padded_cells / kernel / 1e12, 2),
                "peak_Tcells_per_s_full_width": pure, "frac": round(padded_cells / kernel / 1e12 / pure, 4),
                "useful_fraction_of_width": round(float(profile.cells) / padded_cells, 4)}
        line = {
            "metric": "DP cell-updates/s (GCUPS) on 1M-pair Levenshtein batch" if args.config == 2 else f"DP cell-updates/s (GCUPS), {load.name}",
            "value": round(value, 1), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # what the collective layer itself says about the job: backend ("nccl" IS RCCL on ROCm), its rank count, distinct devices
            "ranks": {"backend": dist.get_backend() if distributed else None, "world": dist.get_world_size() if distributed else 1,
                      "devices": devices_in_use},
            "dtype": {0: "u32 bit-vectors (u64 results)", 16: "i16 cells, two per VALU op (i64 results)", 32: "i32 cells (i64 results)",
                      64: "i64 cells"}.get(int(profile.cell_bits), "u32"),
            "data": "synthetic",
            "config": {"workload": load.name, "pairs_per_gpu": rows * columns, "cells_per_gpu": int(profile.cells),
                       "sharding": "query row blocks, candidates replicated" if world > 1 else "single GPU",
                       "entry_point": ENTRY_POINTS[load.kind], "generator": args.generator,
                       # "fresh batches": two different batches of this shape alternate, every timed call plans its tapes afresh
                       "stream": "fresh batches alternate" if len(steps_of) > 1 else "same batch every step"},
            "roofline": line_roofline,
            "host_overhead_ms_per_step": round(ms_per_step - kernel * 1e3, 4),
            "planner": "+".join({0: "host", 1: "device", 2: "device, speculated", 3: "re-used", 4: "inside the scoring launch", 5: "none (tiny tokens)"}[mode] for mode in sorted(planners)),
            "results_checksum": float(checksum),
            # the three figures the reference's own bench prints per engine (bench/similarities.cuh:344-366: bytes passed, operations =
            # cells, inputs processed, and the device-measured "Kernel" line :303-308)
            "reference_style": {"throughput_gb_s": round(float(load.queries.lengths().sum() * columns + load.candidates.lengths().sum() * rows) * world / (elapsed / args.steps) / 1e9, 1),
                                "efficiency_gops_s": round(value, 1), "pairs_per_second": round(rows * columns * world / (elapsed / args.steps), 0),
                                "kernel_gcups": round(float(profile.cells) / kernel / 1e9, 1)},
        }
        if same_tapes is not None:
            line["config"]["same_tapes_gcups"] = same_tapes["value"]
            line["same_tapes"] = same_tapes
        if not args.no_cpu_baseline:  # rank 0, whatever N: the host cores are this box's
            line["cpu_baseline"] = cpu_baseline(load, gpu_matrix, args.cpu_seconds, cell_bits=profile.cell_bits, verify_seconds=args.verify_seconds)
            line["stream_check"] = {"replayed_calls": replayed, "checksum_mismatches": replay_mismatches, "batches": len(steps_of)}
            if len(matrices_of_batches) > 1:  # the OTHER batch of the alternating stream, cell for cell as well
                other_load = type("other", (), {"queries": other_step.keepalive[2], "candidates": other_step.keepalive[3], "kind": load.kind,
                                                "costs": load.costs, "name": load.name + " (the alternate batch)", "table": getattr(load, "table", None)})
                checked = cpu_baseline(other_load, matrices_of_batches[1], 1.0, cell_bits=profile.cell_bits, with_serial=False,
                                       verify_seconds=args.verify_seconds).get("checked", {})
                line["stream_check"]["alternate_batch_whole_matrix"] = bool(checked.get("whole_matrix"))
            clock["headline_cpu_done"] = time.perf_counter()
            attach_cpu_baselines(records, args.extra_cpu_seconds, args.verify_seconds)
        for record in records:
            record.pop("_cpu_baseline_inputs", None), record.pop("_fingerprints_inputs", None)
        clock["done"] = time.perf_counter()
        line["run_seconds"] = {"total": round(clock["done"] - clock["started"], 1), "gpu_legs": round(clock["gpu_done"] - clock["started"], 1),
                               "cpu_baselines": round(clock["done"] - clock["gpu_done"], 1)}
        emit(line, records, args.details)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


# ---- output -----------------------------------------------------------------------------------------------------------------

HEADLINE_LIMIT = 4096  # bytes: round 4's line carried eight config records, grew to 22 KB and the driver kept only its tail


def headline(line, records=()):
    """The LAST stdout line: the headline alone, under HEADLINE_LIMIT bytes whatever the records hold.  Everything the contract
    names stays; of the `configs` records only wall GCUPS per config (`configs_gcups`).  Optional objects are dropped, least
    important first, should the line ever grow past the limit again."""
    short = dict(line)
    short.pop("configs", None)
    summary = {}
    for record in records:
        if "value" in record and "config" in record:
            key = str(record["config"]) + ("@node" if str(record.get("entry_point", "")).startswith("szs_rocm_node") else "")
            summary[key] = record["value"]
        elif "error" in record:
            summary[str(record.get("config"))] = "error"
    if summary:
        short["configs_gcups"] = summary
    for optional in ("reference_style", "same_tapes", "run_seconds", "configs_gcups", "planner", "host_overhead_ms_per_step"):
        if len(json.dumps(short)) < HEADLINE_LIMIT:
            break
        short.pop(optional, None)
    for inner in ("myers_ceiling", "valu"):
        if len(json.dumps(short)) < HEADLINE_LIMIT:
            break
        short["roofline"] = {k: v for k, v in short.get("roofline", {}).items() if k != inner}
    text = json.dumps(short)
    assert len(text) < HEADLINE_LIMIT, f"headline line is {len(text)} bytes"
    return text


def audit(line, records=(), summary=None):
    """What must never leave this program again: a roofline fraction above 1 (the priced work is then not what the
    kernel does), an ESTIMATE where a committed counter pass exists, a headline whose `roofline.kernel` is not the kernel its timed
    leg launches.  Returns the list of problems (empty: clean); `emit` prints it on a line of its own and keeps it in the headline."""
    problems = []
    if summary is None:
        summary, _ = _profile_json("pmc_configs.json")

    def fractions(node, path):
        if isinstance(node, dict):
            for key, value in node.items():
                if key in ("frac", "frac_of_measured_peak") and isinstance(value, (int, float)) and not 0.0 <= value <= 1.0:
                    problems.append(f"{path}.{key} = {value}: outside [0, 1]")
                fractions(value, f"{path}.{key}")

    fractions(line.get("roofline", {}), "headline.roofline")
    for record in records:
        name = f"configs[{record.get('config')}]"
        fractions(record.get("roofline", {}), name + ".roofline")
        counted = str(record.get("roofline", {}).get("counted", ""))
        if "ESTIMATE" in counted and record.get("config") == "fingerprints" and _by_prefix(summary, "cfg11:fingerprint_segments_kernel")[1]:
            problems.append(f"{name}.roofline.counted is an estimate although a committed PMC pass of its kernel exists")
    kernel, planner = str(line.get("roofline", {}).get("kernel", "")), str(line.get("planner", ""))
    if kernel and planner:
        fused_leg = planner == "inside the scoring launch"
        if fused_leg != ("fused" in kernel) and ("myers_short" in kernel):
            problems.append(f"headline.roofline.kernel = {kernel} but the timed leg's planner mode is '{planner}': the counters are another launch's")
    return problems


def emit(line, records, details_path):
    """Prints one `{"configs_record": ...}` line per record, then - LAST - the headline alone; writes both, untrimmed, to
    `details_path` (gpurun_out/bench_configs.json by default; merged back from the GPU box)."""
    problems = audit(line, records)
    line["audit"] = problems
    for record in records:
        print(json.dumps({"configs_record": record}), flush=True)
    if problems:
        print(json.dumps({"audit": problems}), flush=True)
    if details_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(details_path)), exist_ok=True)
            with open(details_path, "w") as handle:
                json.dump({"headline": line, "configs": list(records)}, handle, indent=1)
        except OSError as problem:
            print(json.dumps({"note": f"details file not written: {problem!r}"}), flush=True)
    print(headline(line, records), flush=True)


if __name__ == "__main__":
    main()
