#!/usr/bin/env python3
"""What ONE GPU of an N-GPU node would spend on its share of a config: the rows LPT deals to shard 0 of N, scored here.
    python scripts/measure_shard_of.py --config 5 --shards 1,2,4,8
A one-GPU preview of the strong-scaling curve (replication and the other GPUs' timing jitter aside)."""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")  # the application asks for the wide stream fan-out
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stringzilla_amd as szs
from stringzilla_amd import sharded, workloads
import bench

parser = argparse.ArgumentParser()
parser.add_argument("--config", type=int, default=5)
parser.add_argument("--shards", default="1,2,4,8")
args = parser.parse_args()
load = workloads.config(args.config)
scope = szs.DeviceScope(gpu_device=0)
engine = bench.make_engine(load, scope)
base = None
for shards in [int(x) for x in args.shards.split(",")]:
    shard_of_row, loads = sharded.shard_rows(load.queries.lengths(), shards)
    rows = np.nonzero(shard_of_row == 0)[0]
    queries = load.queries.select(rows).to_device(0)
    candidates = load.candidates.to_device(0)
    out = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device="cuda")
    step = bench.make_step(engine, scope, load, queries, candidates, out, 0)
    wall, kernel, repeats = bench.time_config(step, engine, 2.0, torch.cuda.synchronize)
    cells = int(engine.last_call_profile().cells)
    base = base or wall
    print(json.dumps({"config": args.config, "shards": shards, "rows": len(rows), "ms": round(wall * 1e3, 3), "kernel_ms": round(kernel * 1e3, 3),
                      "gcups_this_gpu": round(cells / wall / 1e9, 1), "speedup_vs_one": round(base / wall, 2), "efficiency": round(base / wall / shards, 3)}), flush=True)
