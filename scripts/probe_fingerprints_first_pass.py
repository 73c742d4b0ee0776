#!/usr/bin/env python3
"""probe_fingerprints_first_pass.py - `szs_fingerprints_u32tape` call by call right after the engine is created: wall time of
every call, to be read beside `rocprofv3 --kernel-trace` of the same run (the kernels' own durations in dispatch order).
bench.py's fingerprints record saw its FIRST pass of calls at twice the time of every later one (profiles/r04/
fingerprints_first_pass.txt); this tells a slow kernel from a slow host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import workloads  # noqa: E402

gpu = szs.DeviceScope(gpu_device=0)
texts = workloads.random_tape(np.random.default_rng(11), 1024, 8192, 12288, workloads.ASCII_PRINTABLE).to_device(0)
engine = szs.Fingerprints(1024, capabilities=gpu)
walls = []
for call in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize()
    started = time.perf_counter()
    engine(texts, device=gpu)
    walls.append((time.perf_counter() - started) * 1e3)
print("wall ms per call:", " ".join(f"{w:.2f}" for w in walls))
