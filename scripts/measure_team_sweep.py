#!/usr/bin/env python3
"""Where the team tier pays: kernel time of square batches over a grid of (query length, candidate length) under the
one-pair-per-lane kernel (`team` = 0) and the compiled team shapes.  Feeds the shape rule of csrc/host/dispatch.c."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from stringzilla_amd import _abi, matrices, workloads

parser = argparse.ArgumentParser()
parser.add_argument("--shapes", default="0,163202,43202")
parser.add_argument("--kinds", default="nw_linear,sw_affine")
parser.add_argument("--queries", default="16,32,64,128,256,512,1024,2048")
parser.add_argument("--candidates", default="64,256,1024")
args = parser.parse_args()
gpu = szs.DeviceScope(gpu_device=0)
rng = np.random.default_rng(3)
_abi.tuning_set("tier", "lanes")
for kind in args.kinds.split(","):
    if kind == "nw_linear":
        engine, alphabet = szs.NeedlemanWunschScores(*matrices.blosum62(), open=-4, extend=-4, capabilities=gpu), workloads.AMINO_ACIDS
    else:
        engine, alphabet = szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu), workloads.NUCLEOTIDES
    for q_length in [int(x) for x in args.queries.split(",")]:
        for c_length in [int(x) for x in args.candidates.split(",")]:
            side = int(min(1024, max(64, (2.0e10 / (q_length * c_length)) ** 0.5)))
            queries = workloads.random_tape(rng, side, q_length * 3 // 4, q_length * 5 // 4, alphabet).to_device(0)
            candidates = workloads.random_tape(rng, side, c_length * 3 // 4, c_length * 5 // 4, alphabet).to_device(0)
            out = torch.empty((side, side), dtype=torch.int64, device="cuda")
            record, reference = {"kind": kind, "queries": q_length, "candidates": c_length, "side": side}, None
            for shape in [int(x) for x in args.shapes.split(",")]:
                _abi.tuning_set("team", shape)
                engine(queries, candidates, device=gpu, out=out)
                times = []
                for _ in range(3):
                    engine(queries, candidates, device=gpu, out=out)
                    times.append(engine.last_call_profile().kernel_milliseconds)
                checksum = int(out.sum().item())
                reference = checksum if reference is None else reference
                assert checksum == reference, (kind, q_length, c_length, shape)
                profile = engine.last_call_profile()
                record[str(shape) if profile.team == shape else f"{shape}->{profile.team}/{profile.cell_bits}"] = round(profile.cells / min(times) / 1e6)  # GCUPS
            print(json.dumps(record), flush=True)
