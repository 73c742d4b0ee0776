#!/usr/bin/env python3
"""The one-launch bit-parallel kernel (hip/myers_queue.hip) against the per-width launches, on a config and on one GPU's share of it:
    python scripts/measure_queue.py --config 5 --shards 1,8 --words auto,4,8,12,16 --rounds auto,1
One JSON line per (shards, setting): wall and kernel milliseconds of the whole C-ABI call, launches, work items, checksum."""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")  # the per-width launches want the wide stream fan-out
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stringzilla_amd as szs
from stringzilla_amd import _abi, sharded, workloads
import bench

parser = argparse.ArgumentParser()
parser.add_argument("--config", type=int, default=5)
parser.add_argument("--shards", default="1,8")
parser.add_argument("--words", default="auto,4,8,12,16")
parser.add_argument("--rounds", default="auto")
parser.add_argument("--priority", default="auto", help="auto | 0 | 1 (comma list): longest chain first on the SIMD, or one priority")
parser.add_argument("--seconds", type=float, default=0.6)
args = parser.parse_args()
load = workloads.config(args.config)
scope = szs.DeviceScope(gpu_device=0)
engine = bench.make_engine(load, scope)
for shards in [int(x) for x in args.shards.split(",")]:
    shard_of_row, loads = sharded.shard_rows(load.queries.lengths(), shards)
    rows = np.nonzero(shard_of_row == 0)[0]
    queries = load.queries.select(rows).to_device(0)
    candidates = load.candidates.to_device(0)
    out = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device="cuda")
    step = bench.make_step(engine, scope, load, queries, candidates, out, 0)
    settings = [("per-width launches", {"queue": 0})]
    for words in args.words.split(","):
        for rounds in args.rounds.split(","):
            for priority in args.priority.split(","):
                settings.append((f"queue words={words} rounds={rounds} priority={priority}",
                                 {"queue": None, "queue_words": None if words == "auto" else words,
                                  "queue_rounds": None if rounds == "auto" else rounds,
                                  "queue_priority": None if priority == "auto" else priority}))
    reference_sum = None
    for name, knobs in settings:
        for knob, value in knobs.items():
            _abi.tuning_set(knob, value)
        out.zero_()
        wall, kernel, repeats = bench.time_config(step, engine, args.seconds, torch.cuda.synchronize)
        profile = engine.last_call_profile()
        checksum = int(out.sum().item())
        reference_sum = checksum if reference_sum is None else reference_sum
        print(json.dumps({"config": args.config, "shards": shards, "rows": len(rows), "setting": name, "ms": round(wall * 1e3, 3),
                          "kernel_ms": round(kernel * 1e3, 3), "launches": int(profile.launches), "queue_items": int(profile.queue_items),
                          "queue_tiles": int(profile.queue_tiles), "tcups": round(int(profile.cells) / wall / 1e12, 2),
                          "same_cells": checksum == reference_sum, "repeats": repeats}), flush=True)
        for knob in knobs:
            _abi.tuning_set(knob, None)
