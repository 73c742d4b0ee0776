#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's config, through the C-ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Metric : DP cell-updates/s (GCUPS) = sum over scored pairs of len(query) * len(candidate) / seconds / 1e9
         (the reference's accounting, /root/reference/bench/similarities.cuh:344-366).
Step   : one `szs_levenshtein_distances_u32tape` call - the whole synchronous C-ABI call, host planning included - over
         config 2: a 1024 x 1024 cross-product (1,048,576 pairs) of printable-ASCII strings, length U[96,160], unit costs.
         Tapes and the results matrix are resident in HBM before the timed region starts.
N > 1  : one process per GPU.  The batch shards by QUERY ROW BLOCKS: every rank scores its own
         1024 query rows against the same 1024 candidates (broadcast once over RCCL/xGMI before timing), so per-GPU work
         is fixed: "weak" scaling, and the timed path has no collective (rows are independent; results stay sharded).
Lines  : rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM with ALGORITHMIC bytes
         (272 B per pair at len 128: len(q) + len(c) + 2 offsets + one 8-byte result) over the
         hipEvent-measured kernel time the library records on its own stream.  `cpu_baseline` times the reference's own
         SIMD engines (oracle/_ref, built from /root/reference) on this box's host cores - a reported baseline only.
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
PROFILES = os.path.join(ROOT, "profiles", "r01")


def _profile_json(name):
    try:
        with open(os.path.join(PROFILES, name)) as handle:
            return json.load(handle)
    except (OSError, ValueError):
        return None


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--config", type=int, default=2, help="BASELINE.json config index (2 = the metric's config)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    parser.add_argument("--same-device", action="store_true",
                        help="testing aid: every rank uses cuda:0 (with --backend gloo), to exercise the N > 1 code "
                             "path on a one-GPU box; the numbers of such a run mean nothing")
    parser.add_argument("--hbm-traffic-bytes", type=float, default=None,
                        help="HBM bytes per launch from a separate rocprofv3 --pmc pass; default: the committed summary "
                             "profiles/r01/pmc_summary.json of this same command (scripts/profile_gpu.sh)")
    return parser.parse_args()


def cpu_baseline(load, gpu_matrix):
    """Times the reference's own CPU engines (best SIMD tier, all host threads) on the same batch and checks that they
    produce the very matrix the GPU produced.  Test-infrastructure code path: the only place bench.py touches oracle/."""
    from oracle import binding

    strings = lambda tape: [tape[i] for i in range(len(tape))]
    queries, candidates = strings(load.queries), strings(load.candidates)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if binding.reference_available():
        try:
            tier = binding.reference_best_tier()
            checker = binding.reference(tier=tier, threads=cores)
            kind, label = "reference", {0: "serial", 1: "haswell (AVX2)", 2: "icelake (AVX-512)"}[tier]
        except OSError:
            checker = None
    else:
        checker = None
    if checker is None:
        checker, kind, label, cores = binding.oracle(), "port", "plain-C oracle", 1
    run = lambda: checker.levenshtein(queries, candidates, **load.costs)
    started = time.perf_counter()
    matrix = run()
    first = time.perf_counter() - started
    assert np.array_equal(matrix, gpu_matrix), "CPU baseline and GPU disagree"
    repeats = int(max(1, min(50, 10.0 / max(first, 1e-3))))  # about 10 s of CPU work in total
    started = time.perf_counter()
    for _ in range(repeats):
        run()
    elapsed = (time.perf_counter() - started) / repeats
    return {
        "value": round(load.cells / elapsed / 1e9, 2), "unit": "GCUPS", "cores": cores, "kind": kind,
        "sample": f"the full {len(queries)}x{len(candidates)} batch of the timed config, {repeats} repeats, "
                  f"{label} tier, {cores} threads, tape packing included; matrix verified equal to the GPU's",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import stringzilla_amd as szs
    from stringzilla_amd import _abi, workloads

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
    load = workloads.config(args.config)
    if load.kind != "levenshtein":
        raise SystemExit("bench.py times the Levenshtein path; other configs are parity-test cases")
    if rank:
        rng = np.random.default_rng(args.config + 1000 * rank)
        low, high = int(load.queries.lengths().min()), int(load.queries.lengths().max())
        load.queries = workloads.random_tape(rng, len(load.queries), 96 if args.config == 2 else low,
                                             160 if args.config == 2 else high, workloads.ASCII_PRINTABLE)
    queries = load.queries.to_device(local_rank)
    if world > 1:  # the one exchange step of the path: replicate the candidates tape over RCCL / xGMI, before timing
        size = torch.tensor([load.candidates.data.size], device=where)
        dist.broadcast(size, 0)
        data = torch.from_numpy(load.candidates.data).to(where) if rank == 0 else torch.empty(int(size), dtype=torch.uint8, device=where)
        offsets = torch.from_numpy(load.candidates.offsets.view(np.int32)).to(where) if rank == 0 else torch.empty(len(load.candidates) + 1, dtype=torch.int32, device=where)
        dist.broadcast(data, 0)
        dist.broadcast(offsets, 0)
        load.candidates = szs.Strs.from_tape(data.cpu().numpy(), offsets.cpu().numpy().view(np.uint32))
        load.candidates._device = (local_rank, data, offsets)
    candidates = load.candidates.to_device(local_rank)

    scope = szs.DeviceScope(gpu_device=local_rank)
    engine = szs.LevenshteinDistances(**load.costs, capabilities=scope)
    rows, columns = len(queries), len(candidates)
    results = torch.empty((rows, columns), dtype=torch.int64, device=where)
    q_tape, c_tape = queries._tape(local_rank), candidates._tape(local_rank)
    error = ctypes.c_char_p()

    def step():
        status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, scope.handle, ctypes.byref(q_tape), ctypes.byref(c_tape),
                                                            results.data_ptr(), columns, ctypes.byref(error))
        if status:
            raise RuntimeError(f"szs_levenshtein_distances_u32tape failed: {status} {error.value}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    fence()
    started = time.perf_counter()
    for _ in range(args.steps):
        step()  # synchronous: returns after the scope's stream has drained
        kernel_ms.append(engine.last_call_profile().kernel_milliseconds)
    fence()
    elapsed = time.perf_counter() - started
    if world > 1:
        slowest = torch.tensor([elapsed], dtype=torch.float64, device=where)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        elapsed = float(slowest)

    profile = engine.last_call_profile()
    cells_per_rank = torch.tensor([float(profile.cells)], dtype=torch.float64, device=where)
    checksum = results.sum().reshape(1).to(torch.float64)
    if world > 1:
        dist.all_reduce(cells_per_rank)
        dist.all_reduce(checksum)
    total_cells = float(cells_per_rank)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells * args.steps / elapsed / 1e9
        kernel = float(np.mean(kernel_ms)) * 1e-3  # seconds per launch group, hipEvent pair on the library's stream
        achieved = profile.algorithmic_bytes / kernel / 1e9
        gpu_matrix = results.cpu().numpy().view(np.uint64)
        # HBM traffic of the dominant kernel, per launch: FETCH_SIZE + WRITE_SIZE from their own rocprofv3 --pmc passes
        # of this command (scripts/profile_gpu.sh), kilobyte units and the gfx950 wide-read correction applied by
        # scripts/pmc_summary.py as MI355X_MICROARCH.md prescribes.  Committed, not collected inside this process.
        traffic = args.hbm_traffic_bytes
        traffic_note = "from --hbm-traffic-bytes" if traffic is not None else "no committed PMC pass for this config"
        if traffic is None and args.config == 2:
            summary = next((counters for name, counters in (_profile_json("pmc_summary.json") or {}).items()
                            if name.startswith("levenshtein_myers_short_kernel")), None)  # the name carries template arguments
            if summary and "hbm_fetch_bytes_raw" in summary and "hbm_write_bytes_raw" in summary:
                traffic = summary["hbm_fetch_bytes_raw"] + summary["hbm_write_bytes_raw"]
                traffic_note = ("profiles/r01/pmc_summary.json: FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch (raw; the x2 "
                                "wide-stream read correction would give %d)" % int(
                                    summary["hbm_fetch_bytes_x2_wide_stream_correction"] + summary["hbm_write_bytes_raw"]))
        valu = (_profile_json("valu_peak.json") or {})
        pure = valu.get("myers_pure_W4_Tcells")
        lengths = load.queries.lengths().astype(np.int64)
        padded_cells = float((np.maximum(1, -(-lengths // 32)) * 32).sum()) * float(load.candidates.lengths().sum())
        line = {
            "metric": "DP cell-updates/s (GCUPS) on 1M-pair Levenshtein batch", "value": round(value, 1), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 bit-vectors (u64 results)",
            "data": "synthetic",
            "config": {"workload": load.name, "pairs_per_gpu": rows * columns, "cells_per_gpu": int(profile.cells),
                       "sharding": "query row blocks, candidates replicated" if world > 1 else "single GPU",
                       "entry_point": "szs_levenshtein_distances_u32tape"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": ("levenshtein_myers_short_kernel (1 launch per step)" if profile.launches == 1 else
                                    "levenshtein_myers_short_kernel + long-width kernels (%d launches per step)" % profile.launches),
                         "kernel_ms": round(kernel * 1e3, 4),
                         "algorithmic_bytes": int(profile.algorithmic_bytes),
                         "kernel_gcups": round(profile.cells / kernel / 1e9, 1),
                         "note": "integer-VALU bound by construction (HBM traffic is ~4% of the algorithmic bytes: tapes "
                                 "are L2-resident); the HBM fraction is reported because the metric asks for it, the "
                                 "`valu` object is the roofline that says something about the kernel",
                         "valu": {
                             "bound": "integer VALU issue, measured",
                             "achieved_Tcells_per_s_full_width": round(padded_cells / kernel / 1e12, 2),
                             "peak_Tcells_per_s_full_width": pure,
                             "frac": round(padded_cells / kernel / 1e12 / pure, 4) if pure else None,
                             "peak_source": "scripts/valu_peak.hip `myers_pure_W4_Tcells`: the kernel's column update on "
                                            "register-resident masks (profiles/r01/valu_peak.json)",
                             "useful_fraction_of_width": round(float(profile.cells) / padded_cells, 4)}},
            "host_overhead_ms_per_step": round(ms_per_step - kernel * 1e3, 4),
            "results_checksum": float(checksum),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(load, gpu_matrix)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
