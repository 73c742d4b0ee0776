#!/usr/bin/env python3
"""measure_team_fixed.py - the team tier's shapes on DNA batches of FIXED lengths (Smith-Waterman, NUC.4.4, affine -4 / -1): what is left of
the difference between sixteen-lane and wave-wide teams when no row is padded and every pass is whole."""
import os as _os; _os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from stringzilla_amd import _abi, matrices, workloads

gpu = szs.DeviceScope(gpu_device=0)
engine = szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu)
_abi.tuning_set("tier", "lanes")
dna = np.frombuffer(b"ACGT", dtype=np.uint8)
for rows, q_len, c_len in ((512, 4096, 4096), (512, 2048, 4096), (512, 4096, 1024), (512, 4000, 4096)):
    rng = np.random.default_rng(rows + q_len)
    queries = workloads.random_tape(rng, rows, q_len, q_len, dna).to_device(0)
    candidates = workloads.random_tape(rng, rows, c_len, c_len, dna).to_device(0)
    out = torch.empty((rows, rows), dtype=torch.int64, device="cuda")
    reference = None
    for shape in (163202, 643202):
        _abi.tuning_set("team", shape)
        engine(queries, candidates, device=gpu, out=out)
        kernel = []
        for _ in range(2):
            engine(queries, candidates, device=gpu, out=out)
            kernel.append(engine.last_call_profile().kernel_milliseconds)
        profile = engine.last_call_profile()
        checksum = int(out.sum().item())
        reference = checksum if reference is None else reference
        print(json.dumps({"rows": rows, "query": q_len, "candidate": c_len, "team": profile.team, "kernel_ms": round(min(kernel), 3),
                          "gcups": round(profile.cells / min(kernel) / 1e6, 1), "agrees": checksum == reference}), flush=True)
