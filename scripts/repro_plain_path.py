#!/usr/bin/env python3
"""repro_plain_path.py [ROUNDS] - the set-up of tests/test_gpu_round6.py::_soak again and again: a fresh engine, `fused` knob 0, three batches of
1024 x 1024 strings of 96 ... 160 bytes through the raw C-ABI into zeroed matrices, every matrix compared with the reference's engines.
Prints one JSON line per mismatch (where, how many, what the call's profile said) and a summary."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from oracle import binding
from stringzilla_amd import _abi, workloads

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sync_fills = len(sys.argv) > 2 and sys.argv[2] == "sync"  # `repro_plain_path.py 30 sync`: wait for the zero fill before every call
gpu = szs.DeviceScope(gpu_device=0)
threads = len(os.sched_getaffinity(0))
checker = binding.reference(tier=binding.reference_best_tier(), threads=threads)
side = 1024
tapes = [(workloads.random_tape(np.random.default_rng(100 + 2 * b), side, 96, 160, workloads.ASCII_PRINTABLE).to_device(0),
          workloads.random_tape(np.random.default_rng(101 + 2 * b), side, 96, 160, workloads.ASCII_PRINTABLE).to_device(0)) for b in range(3)]
truths = [checker.levenshtein([q[i] for i in range(side)], [c[i] for i in range(side)]) for q, c in tapes]
failures = 0
for knobs in ({"fused": 0}, {"fused": 0, "speculate": 0}, {}):
    for name, value in knobs.items():
        _abi.tuning_set(name, value)
    for round_index in range(rounds):
        engine = szs.LevenshteinDistances(capabilities=gpu)
        for b, (queries, candidates) in enumerate(tapes):
            out = torch.zeros((side, side), dtype=torch.int64, device="cuda:0")
            if sync_fills:
                torch.cuda.synchronize()  # the fill runs on TORCH's stream, the scoring launch on the scope's: nothing else orders them
            q_tape, c_tape, error = queries._tape(0), candidates._tape(0), ctypes.c_char_p()
            status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(q_tape), ctypes.byref(c_tape), out.data_ptr(), side,
                                                                ctypes.byref(error))
            profile = engine.last_call_profile()
            got = out.cpu().numpy().view(np.uint64)
            wrong = np.argwhere(got != truths[b])
            if status or len(wrong):
                failures += 1
                print(json.dumps({"knobs": knobs, "round": round_index, "batch": b, "status": status, "wrong_cells": int(len(wrong)),
                                  "planner": int(profile.planner), "launches": int(profile.launches), "first": wrong[:6].tolist(),
                                  "got": [int(got[tuple(w)]) for w in wrong[:6]], "expected": [int(truths[b][tuple(w)]) for w in wrong[:6]],
                                  "rows_hit": len(set(wrong[:, 0].tolist())), "columns_hit": sorted(set(wrong[:, 1].tolist()))[:10]}), flush=True)
    for name in knobs:
        _abi.tuning_set(name, None)
print(json.dumps({"rounds_per_setting": rounds, "sync_fills": sync_fills, "failures": failures}))
