#!/usr/bin/env python3
"""Raw throughput of the bodies of hip/myers_queue.hip against the per-width kernels of hip/lev_myers.hip on batches of ONE length
class (so that nothing but the body and the item overhead differs): `queue` knob 1 vs 0, words per lane pinned.
    python scripts/measure_queue_shapes.py [--seconds 0.4]"""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stringzilla_amd as szs
from stringzilla_amd import _abi, workloads
import bench

parser = argparse.ArgumentParser()
parser.add_argument("--seconds", type=float, default=0.4)
parser.add_argument("--shapes", default="1024x128,1024x300,1024x500,1024x1000,512x2000,3000x40")
args = parser.parse_args()
scope = szs.DeviceScope(gpu_device=0)
engine = szs.LevenshteinDistances(capabilities=scope)
load = workloads.config(2, scale=1 / 64)  # only for make_step's entry point
for shape in args.shapes.split(","):
    count, length = (int(x) for x in shape.split("x"))
    rng = np.random.default_rng(count + length)
    low, high = max(1, length - length // 4), length + length // 4
    queries = workloads.random_tape(rng, count, low, high, workloads.ASCII_PRINTABLE).to_device(0)
    candidates = workloads.random_tape(rng, count, low, high, workloads.ASCII_PRINTABLE).to_device(0)
    out = torch.empty((count, count), dtype=torch.int64, device="cuda")
    step = bench.make_step(engine, scope, load, queries, candidates, out, 0)
    settings = [("per-width kernels", {"queue": 0})] + [(f"queue words={w}", {"queue": 1, "queue_words": None if w == "auto" else w})
                                                        for w in ("auto", 4, 8, 12, 16)]
    reference_sum = None
    for name, knobs in settings:
        for knob, value in knobs.items():
            _abi.tuning_set(knob, value)
        out.zero_()
        wall, kernel, repeats = bench.time_config(step, engine, args.seconds, torch.cuda.synchronize)
        profile = engine.last_call_profile()
        checksum = int(out.sum().item())
        reference_sum = checksum if reference_sum is None else reference_sum
        print(json.dumps({"shape": shape, "setting": name, "kernel_ms": round(kernel * 1e3, 3), "ms": round(wall * 1e3, 3),
                          "kernel_tcups": round(int(profile.cells) / kernel / 1e12, 2), "launches": int(profile.launches),
                          "queue_items": int(profile.queue_items), "same_cells": checksum == reference_sum}), flush=True)
        for knob in knobs:
            _abi.tuning_set(knob, None)
