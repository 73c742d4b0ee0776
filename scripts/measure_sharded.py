#!/usr/bin/env python3
"""BASELINE.json configs 4 and 5 the way they are specified: ONE batch, its query rows dealt over the GPUs of a node
(`stringzilla_amd/sharded.py`: LPT on the row lengths, candidates broadcast over RCCL / xGMI, no collective on the data
path), with what the configs ask to be reported - per-GPU busy time, imbalance = max / mean, aggregate GCUPS.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
        scripts/measure_sharded.py --config 5

`bench.py` is the judged line (config 2, weak scaling); this is the strong-scaling companion for the two sharded configs.
`--backend gloo --same-device` runs every rank on cuda:0 - it exercises the code path on a one-GPU box, its numbers mean
nothing."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import stringzilla_amd as szs
from stringzilla_amd import matrices, sharded, workloads

parser = argparse.ArgumentParser()
parser.add_argument("--config", type=int, default=5)
parser.add_argument("--scale", type=float, default=1.0)
parser.add_argument("--repeats", type=int, default=3)
parser.add_argument("--backend", default="nccl")
parser.add_argument("--same-device", action="store_true")
args = parser.parse_args()

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
if args.backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
else:
    dist.init_process_group(args.backend)

load = workloads.config(args.config, scale=args.scale) if rank == 0 else None  # only the source rank holds the batch
kind = [load.kind if load else "", load.table if load and load.table else "", json.dumps(load.costs) if load else ""]
dist.broadcast_object_list(kind, 0)
scope = szs.DeviceScope(gpu_device=local_rank)
if kind[0] == "levenshtein":
    engine = szs.LevenshteinDistances(**json.loads(kind[2]), capabilities=scope)
elif kind[0] == "levenshtein_utf8":
    engine = szs.LevenshteinDistancesUTF8(**json.loads(kind[2]), capabilities=scope)
else:
    cls = szs.NeedlemanWunschScores if kind[0] == "needleman_wunsch" else szs.SmithWatermanScores
    engine = cls(*matrices.by_name(kind[1]), **json.loads(kind[2]), capabilities=scope)

busy = []
def score(queries, candidates):  # this rank's rows x all candidates, on this rank's GPU
    queries.to_device(local_rank), candidates.to_device(local_rank)
    out = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device=torch.device("cuda", local_rank))
    engine(queries, candidates, device=scope, out=out)  # warm-up: allocations, code objects
    torch.cuda.synchronize()
    started = time.perf_counter()
    for _ in range(args.repeats):
        engine(queries, candidates, device=scope, out=out)
    busy.append((time.perf_counter() - started) / args.repeats)
    return out.cpu().numpy()

node = sharded.ShardedEngine(engine=engine, scope=scope, score=score)
rows, local = node(load.queries if load else None, load.candidates if load else None, source=0)
profile = engine.last_call_profile()
mine = torch.tensor([busy[0] if busy else 0.0, float(profile.cells) if busy else 0.0, float(len(rows)), float(local.sum())],
                    dtype=torch.float64, device=torch.device("cuda", local_rank) if args.backend == "nccl" else "cpu")
gathered = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(gathered, mine)
if rank == 0:
    seconds = np.array([float(g[0]) for g in gathered])
    cells = sum(float(g[1]) for g in gathered)
    print(json.dumps({"config": load.name, "n_gpus": world, "cells": int(cells), "rows_per_gpu": [int(g[2]) for g in gathered],
                      "busy_ms_per_gpu": [round(s * 1e3, 3) for s in seconds.tolist()],
                      "imbalance_max_over_mean": round(float(seconds.max() / max(seconds.mean(), 1e-12)), 4),
                      "row_weight_imbalance": round(node.last_balance, 4),
                      "aggregate_gcups": round(cells / seconds.max() / 1e9, 1), "checksum": int(sum(float(g[3]) for g in gathered)),
                      "same_device": bool(args.same_device)}))
dist.barrier()
dist.destroy_process_group()
