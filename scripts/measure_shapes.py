#!/usr/bin/env python3
"""Times the shapes of the reference's own sweep that are NOT a square million-pair batch (bench/similarities.cuh:191-244:
retrieval corner 1 x N, skewed N x 1, small squares of long strings, a single very long pair), once with the planner
free to choose tier and orientation and once pinned to the caller's orientation on the lanes tier - the design this
repository started from.  Prints one JSON line per (shape, engine, mode).  A working tool, not the bench line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from stringzilla_amd import _abi, matrices, workloads

parser = argparse.ArgumentParser()
parser.add_argument("--repeats", type=int, default=3)
parser.add_argument("--lanes-budget", type=float, default=2e10, help="skip the pinned lanes-tier run above this many single-lane cells")
parser.add_argument("--only", default="")
args = parser.parse_args()

gpu = szs.DeviceScope(gpu_device=0)
rng = np.random.default_rng(7)
DNA, PROTEIN, ASCII = workloads.NUCLEOTIDES, workloads.AMINO_ACIDS, workloads.ASCII_PRINTABLE


def engines():
    yield "lev_unit", lambda: szs.LevenshteinDistances(capabilities=gpu), DNA
    yield "nw_blosum62_linear", lambda: szs.NeedlemanWunschScores(*matrices.blosum62(), open=-4, extend=-4, capabilities=gpu), PROTEIN
    yield "sw_nuc44_affine", lambda: szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu), DNA


SHAPES = [  # (label, queries, candidates, length)
    ("1x1 len100000", 1, 1, 100000),
    ("1x1 len10000", 1, 1, 10000),
    ("1x1 len1000", 1, 1, 1000),
    ("16x16 len4096", 16, 16, 4096),
    ("64x64 len512", 64, 64, 512),
    ("128x128 len1000", 128, 128, 1000),
    ("1x4096 len128", 1, 4096, 128),
    ("4096x1 len128", 4096, 1, 128),
    ("32768x8 len128", 32768, 8, 128),
]


def timed(engine, queries, candidates, out, repeats):
    engine(queries, candidates, device=gpu, out=out)
    kernel, wall = [], []
    for _ in range(repeats):
        torch.cuda.synchronize()
        started = time.perf_counter()
        engine(queries, candidates, device=gpu, out=out)
        wall.append(time.perf_counter() - started)
        kernel.append(engine.last_call_profile().kernel_milliseconds * 1e-3)
    return min(kernel), min(wall), engine.last_call_profile()


for label, q_count, c_count, length in SHAPES:
    if args.only and args.only not in label:
        continue
    for name, make, alphabet in engines():
        queries = workloads.random_tape(rng, q_count, length * 3 // 4, length * 5 // 4, alphabet).to_device(0)
        candidates = workloads.random_tape(rng, c_count, length * 3 // 4, length * 5 // 4, alphabet).to_device(0)
        out = torch.empty((q_count, c_count), dtype=torch.int64, device="cuda")
        sums = {}
        for mode in ("auto", "pinned"):
            longest_pair = int(queries.lengths().max()) * int(candidates.lengths().max())
            if mode == "pinned" and longest_pair * (1 if name == "lev_unit" and length <= 2048 else 12) > args.lanes_budget:
                continue  # one lane would walk this pair for seconds to minutes
            if mode == "pinned":
                _abi.tuning_set("tier", "lanes"), _abi.tuning_set("swap", "0")
            else:
                _abi.tuning_set("tier", None), _abi.tuning_set("swap", None)
            engine = make()
            kernel, wall, profile = timed(engine, queries, candidates, out, args.repeats)
            sums[mode] = int(out.sum().item())
            print(json.dumps({
                "shape": label, "engine": name, "mode": mode, "tier": int(profile.tier), "transposed": int(profile.transposed),
                "cells": int(profile.cells), "kernel_ms": round(kernel * 1e3, 3), "wall_ms": round(wall * 1e3, 3),
                "kernel_gcups": round(profile.cells / kernel / 1e9, 1), "wall_gcups": round(profile.cells / wall / 1e9, 1),
                "checksum": sums[mode],
            }), flush=True)
        if len(sums) == 2:
            assert sums["auto"] == sums["pinned"], (label, name, sums)
_abi.tuning_set("tier", None), _abi.tuning_set("swap", None)
