#!/usr/bin/env python3
"""measure_fused.py - the short launch that plans itself (hip/lev_myers.hip: fused kernel) on config 2's shape, call by call:
run with SZS_ROCM_TRACE=1 to see the host phases and the sorting workgroups' own timestamps on stderr."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import workloads  # noqa: E402

gpu = szs.DeviceScope(gpu_device=0)
engine = szs.LevenshteinDistances(capabilities=gpu)
batches = [(workloads.mt19937_64_tape(10 + k, 1024, 96, 160, workloads.ASCII_PRINTABLE).to_device(0),
            workloads.mt19937_64_tape(20 + k, 1024, 96, 160, workloads.ASCII_PRINTABLE).to_device(0)) for k in range(2)]
out = torch.empty((1024, 1024), dtype=torch.int64, device="cuda:0")
for call in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    started = time.perf_counter()
    engine(*batches[call % 2], device=gpu, out=out)
    profile = engine.last_call_profile()
    print(f"call {call}: wall {1e6 * (time.perf_counter() - started):.1f} us, kernel {profile.kernel_milliseconds * 1e3:.1f} us, planner {profile.planner}", file=sys.stderr)
