#!/usr/bin/env python3
"""Per-call kernel and wall time of config 2, call by call from a cold start: shows the clock ramp of short calls."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stringzilla_amd as szs
from stringzilla_amd import workloads
import bench

load = workloads.config(2)
scope = szs.DeviceScope(gpu_device=0)
engine = bench.make_engine(load, scope)
q, c = load.queries.to_device(0), load.candidates.to_device(0)
results = torch.empty((len(q), len(c)), dtype=torch.int64, device="cuda")
step = bench.make_step(engine, scope, load, q, c, results, 0)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE_SECONDS", "0.5")))
kernel, wall = [], []
for _ in range(int(os.environ.get("CALLS", "400"))):
    t = time.perf_counter(); step(); wall.append((time.perf_counter() - t) * 1e6)
    kernel.append(engine.last_call_profile().kernel_milliseconds * 1e3)
k = np.array(kernel); w = np.array(wall)
for lo, hi in ((0, 5), (5, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 400)):
    print(f"calls {lo:3d}-{hi:3d}: kernel {k[lo:hi].mean():7.1f} us  wall {w[lo:hi].mean():7.1f} us")

def series(label, calls=60):
    kernel = []
    for _ in range(calls):
        step(); kernel.append(engine.last_call_profile().kernel_milliseconds * 1e3)
    k = np.array(kernel)
    print(f"{label}: first 5 {k[:5].mean():.1f}  5-25 {k[5:25].mean():.1f}  25-60 {k[25:].mean():.1f}")

a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for gap in (0.0, 0.001, 0.005, 0.02, 0.1):
    for _ in range(30): (a @ a)
    torch.cuda.synchronize(); time.sleep(gap)
    series(f"after ~100 ms of bf16 GEMMs + {gap * 1e3:.0f} ms idle")
for gap in (0.0, 0.001, 0.005, 0.02, 0.1):
    for _ in range(300): step()
    time.sleep(gap)
    series(f"after 300 own calls + {gap * 1e3:.0f} ms idle")
