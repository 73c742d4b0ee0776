#!/usr/bin/env python3
"""Times one BASELINE.json config (or a share of its query rows) under every compiled shape of the team tier and under the
one-pair-per-lane kernel (`team` = 0).  A working tool for profiles/ and the shape table in DESIGN.md."""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")  # the application asks for the wide stream fan-out
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import stringzilla_amd as szs
from stringzilla_amd import _abi, matrices, workloads

parser = argparse.ArgumentParser()
parser.add_argument("--config", type=int, default=4)
parser.add_argument("--scale", type=float, default=1.0)
parser.add_argument("--repeats", type=int, default=2)
parser.add_argument("--shapes", default="")
parser.add_argument("--shards", type=int, default=1, help="score only the query rows LPT deals to GPU 0 of N")
parser.add_argument("--tier", default="lanes")
args = parser.parse_args()

gpu = szs.DeviceScope(gpu_device=0)
load = workloads.config(args.config, scale=args.scale)
if load.kind == "levenshtein":  # configs 7 / 8: non-unit costs, the team tier's `distance` objective
    engine = szs.LevenshteinDistances(**load.costs, capabilities=gpu)
else:
    cls = szs.NeedlemanWunschScores if load.kind == "needleman_wunsch" else szs.SmithWatermanScores
    engine = cls(*matrices.by_name(load.table), **load.costs, capabilities=gpu)
import numpy as np
from stringzilla_amd import sharded
shard_of_row, _ = sharded.shard_rows(load.queries.lengths(), args.shards)
queries = load.queries.select(np.nonzero(shard_of_row == 0)[0]).to_device(0)
load.candidates.to_device(0)
if args.tier != "auto":
    _abi.tuning_set("tier", args.tier)
out = torch.empty((len(queries), len(load.candidates)), dtype=torch.int64, device="cuda")
shapes = [int(x) for x in args.shapes.split(",")] if args.shapes else [0] + _abi.team_shapes()
reference = None
for shape in shapes:
    _abi.tuning_set("team", shape)
    engine(queries, load.candidates, device=gpu, out=out)  # warm-up
    kernel = []
    for _ in range(args.repeats):
        engine(queries, load.candidates, device=gpu, out=out)
        kernel.append(engine.last_call_profile().kernel_milliseconds)
    profile = engine.last_call_profile()
    checksum = int(out.sum().item())
    reference = checksum if reference is None else reference
    print(json.dumps({"config": load.name, "rows": len(queries), "tier": profile.tier, "team": profile.team, "cell_bits": profile.cell_bits,
                      "kernel_ms": round(min(kernel), 3), "gcups": round(profile.cells / min(kernel) / 1e6, 1),
                      "checksum": checksum, "agrees": checksum == reference}), flush=True)
