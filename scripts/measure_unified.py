#!/usr/bin/env python3
"""Config 2 with the tapes and the results in UNIFIED memory from the library's own allocator (`szs_unified_alloc`,
what the reference's Python binding hands to a GPU engine: python/stringzillas/similarities.c:268-272) against the same
call on plain device memory.  Prints one JSON line per placement."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import stringzilla_amd as szs
from stringzilla_amd import _abi, workloads

gpu = szs.DeviceScope(gpu_device=0)
load = workloads.config(2)
engine = szs.LevenshteinDistances(capabilities=gpu)
rows, columns = len(load.queries), len(load.candidates)


def unified_copy(array):
    pointer = _abi.lib.szs_unified_alloc(max(array.nbytes, 1))
    ctypes.memmove(pointer, array.ctypes.data, array.nbytes)
    return pointer


def run(q_struct, c_struct, results_pointer, repeats=20):
    error = ctypes.c_char_p()
    best_wall, best_kernel = 1e9, 1e9
    for _ in range(repeats + 2):
        torch.cuda.synchronize()
        started = time.perf_counter()
        status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(q_struct),
                                                            ctypes.byref(c_struct), results_pointer, columns, ctypes.byref(error))
        wall = time.perf_counter() - started
        assert status == 0, error.value
        best_wall, best_kernel = min(best_wall, wall), min(best_kernel, engine.last_call_profile().kernel_milliseconds * 1e-3)
    return best_kernel, best_wall


device_out = torch.empty((rows, columns), dtype=torch.int64, device="cuda")
kernel, wall = run(load.queries._tape(0), load.candidates._tape(0), device_out.data_ptr())
cells = engine.last_call_profile().cells
reference_sum = int(device_out.sum().item())
print(json.dumps({"placement": "device tapes, device results", "kernel_ms": round(kernel * 1e3, 3), "wall_ms": round(wall * 1e3, 3),
                  "wall_gcups": round(cells / wall / 1e9, 1)}), flush=True)

q_struct = _abi.U32Tape(unified_copy(load.queries.data), unified_copy(load.queries.offsets), rows)
c_struct = _abi.U32Tape(unified_copy(load.candidates.data), unified_copy(load.candidates.offsets), columns)
kernel, wall = run(q_struct, c_struct, device_out.data_ptr())
assert int(device_out.sum().item()) == reference_sum
print(json.dumps({"placement": "unified tapes, device results", "kernel_ms": round(kernel * 1e3, 3), "wall_ms": round(wall * 1e3, 3),
                  "wall_gcups": round(cells / wall / 1e9, 1)}), flush=True)

unified_results = _abi.lib.szs_unified_alloc(rows * columns * 8)
kernel, wall = run(q_struct, c_struct, unified_results)
view = np.ctypeslib.as_array(ctypes.cast(unified_results, ctypes.POINTER(ctypes.c_int64)), shape=(rows, columns))
assert int(view.sum()) == reference_sum
print(json.dumps({"placement": "unified tapes, unified results", "kernel_ms": round(kernel * 1e3, 3), "wall_ms": round(wall * 1e3, 3),
                  "wall_gcups": round(cells / wall / 1e9, 1)}), flush=True)

host_out = np.zeros((rows, columns), dtype=np.int64)
kernel, wall = run(load.queries._tape(0), load.candidates._tape(0), host_out.ctypes.data, repeats=5)
assert int(host_out.sum()) == reference_sum
print(json.dumps({"placement": "device tapes, pageable host results (staged)", "kernel_ms": round(kernel * 1e3, 3),
                  "wall_ms": round(wall * 1e3, 3), "wall_gcups": round(cells / wall / 1e9, 1)}), flush=True)
