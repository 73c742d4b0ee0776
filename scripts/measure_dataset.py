#!/usr/bin/env python3
"""Scores a REAL-TEXT dataset the way the reference's benchmark does (`bench/similarities.cuh`, `bench/shared.hpp`):
tokenise the file (`STRINGWARS_DATASET`, `STRINGWARS_TOKENS`, `STRINGWARS_MAX_TOKENS`, `STRINGWARS_UNIQUE`,
`STRINGWARS_SEED` - the reference's own variable names), draw Q x C tokens, score the cross-product on one GPU and
report kernel / wall GCUPS.  There are no datasets in this image; this is the hook for a box that has them."""
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
parser.add_argument("--dataset", default=os.environ.get("STRINGWARS_DATASET"))
parser.add_argument("--tokens", default=os.environ.get("STRINGWARS_TOKENS", "words"))
parser.add_argument("--max-tokens", type=int, default=int(os.environ.get("STRINGWARS_MAX_TOKENS", "0")))
parser.add_argument("--unique", action="store_true", default=os.environ.get("STRINGWARS_UNIQUE", "0") in ("1", "true"))
parser.add_argument("--seed", type=int, default=int(os.environ.get("STRINGWARS_SEED", "0")))
parser.add_argument("--queries", type=int, default=1024)
parser.add_argument("--candidates", type=int, default=1024)
parser.add_argument("--engine", default="levenshtein", choices=["levenshtein", "levenshtein_utf8", "nw_blosum62", "sw_nuc44"])
parser.add_argument("--repeats", type=int, default=5)
args = parser.parse_args()
if not args.dataset:
    sys.exit("--dataset (or STRINGWARS_DATASET) is required")

tokens = workloads.tokenize_dataset(open(args.dataset, "rb").read(), args.tokens, args.max_tokens, args.unique)
rng = np.random.default_rng(args.seed)
pick = lambda count: szs.Strs([tokens[int(i)] for i in rng.integers(0, len(tokens), size=count)])
queries, candidates = pick(args.queries), pick(args.candidates)
gpu = szs.DeviceScope(gpu_device=0)
engine = {"levenshtein": lambda: szs.LevenshteinDistances(capabilities=gpu),
          "levenshtein_utf8": lambda: szs.LevenshteinDistancesUTF8(capabilities=gpu),
          "nw_blosum62": lambda: szs.NeedlemanWunschScores(*matrices.blosum62(), open=-4, extend=-4, capabilities=gpu),
          "sw_nuc44": lambda: szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu)}[args.engine]()
queries.to_device(0), candidates.to_device(0)
out = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device="cuda")
engine(queries, candidates, device=gpu, out=out)
kernel, wall = [], []
for _ in range(args.repeats):
    torch.cuda.synchronize()
    started = time.perf_counter()
    engine(queries, candidates, device=gpu, out=out)
    wall.append(time.perf_counter() - started)
    kernel.append(engine.last_call_profile().kernel_milliseconds * 1e-3)
profile = engine.last_call_profile()
print(json.dumps({"dataset": os.path.basename(args.dataset), "tokens": args.tokens, "tokens_found": len(tokens),
                  "engine": args.engine, "pairs": profile.pairs, "cells": profile.cells,
                  "kernel_gcups": round(profile.cells / min(kernel) / 1e9, 1), "wall_gcups": round(profile.cells / min(wall) / 1e9, 1),
                  "tier": profile.tier, "transposed": profile.transposed, "launches": profile.launches, "planner": profile.planner,
                  "kernel_us": round(min(kernel) * 1e6, 1), "wall_us": round(min(wall) * 1e6, 1),
                  "longest": [int(queries.lengths().max()), int(candidates.lengths().max())],
                  "results_gb_s": round(profile.pairs * 8 / min(kernel) / 1e9, 1)}))
