#!/usr/bin/env python3
"""Times the BASELINE.json configs at full size on one GPU (kernel = hipEvent pair inside the library, wall = whole
C-ABI call).  Not the bench line - a working tool for the per-config table in DESIGN.md / profiles/."""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")  # the application asks for the wide stream fan-out
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from stringzilla_amd import matrices, workloads

parser = argparse.ArgumentParser()
parser.add_argument("--configs", default="1,2,3,4,5")
parser.add_argument("--scale", type=float, default=1.0)
parser.add_argument("--repeats", type=int, default=5)
args = parser.parse_args()

gpu = szs.DeviceScope(gpu_device=0)
for index in [int(x) for x in args.configs.split(",")]:
    load = workloads.config(index, scale=args.scale)
    if load.kind == "levenshtein":
        engine = szs.LevenshteinDistances(**load.costs, capabilities=gpu)
    elif load.kind == "levenshtein_utf8":
        engine = szs.LevenshteinDistancesUTF8(**load.costs, capabilities=gpu)
    else:
        cls = szs.NeedlemanWunschScores if load.kind == "needleman_wunsch" else szs.SmithWatermanScores
        engine = cls(*matrices.by_name(load.table), **load.costs, capabilities=gpu)
    load.queries.to_device(0), load.candidates.to_device(0)
    out = torch.empty((len(load.queries), len(load.candidates)), dtype=torch.int64, device="cuda")
    engine(load.queries, load.candidates, device=gpu, out=out)  # warm-up (allocations, code object load)
    kernel, wall = [], []
    for _ in range(args.repeats):
        torch.cuda.synchronize()
        started = time.perf_counter()
        engine(load.queries, load.candidates, device=gpu, out=out)
        wall.append(time.perf_counter() - started)
        kernel.append(engine.last_call_profile().kernel_milliseconds * 1e-3)
    profile = engine.last_call_profile()
    print(json.dumps({
        "config": load.name, "pairs": int(profile.pairs), "cells": int(profile.cells),
        "kernel_ms": round(min(kernel) * 1e3, 3), "wall_ms": round(min(wall) * 1e3, 3),
        "kernel_gcups": round(profile.cells / min(kernel) / 1e9, 1), "wall_gcups": round(profile.cells / min(wall) / 1e9, 1),
        "algorithmic_GBps": round(profile.algorithmic_bytes / min(kernel) / 1e9, 1), "launches": profile.launches,
        "checksum": int(out.sum().item()),
    }), flush=True)
