#!/usr/bin/env python3
"""Debug aid: systolic tier against the lanes tier, cell by cell, on shapes that separate hypotheses."""
import json, os, sys, time, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stringzilla_amd as szs
from stringzilla_amd import matrices, workloads

faulthandler.dump_traceback_later(230, exit=True)
print('imports done', flush=True)
gpu = szs.DeviceScope(gpu_device=0)
DNA, PROTEIN = workloads.NUCLEOTIDES, workloads.AMINO_ACIDS


def both(make, queries, candidates, label, repeats=3):
    out = {}
    for tier in ("lanes", "systolic"):
        os.environ["SZS_ROCM_TIER"], os.environ["SZS_ROCM_SWAP"] = tier, "0"
        engine = make()
        runs = []
        for _ in range(repeats if tier == "systolic" else 1):
            started = time.perf_counter()
            runs.append(engine(queries, candidates, device=gpu).copy())
            elapsed = time.perf_counter() - started
        out[tier] = runs
    reference = out["lanes"][0]
    report = {"label": label, "last_ms": round(elapsed * 1e3, 2)}
    for i, run in enumerate(out["systolic"]):
        bad = np.argwhere(run != reference)
        report[f"run{i}_bad"] = len(bad)
        if len(bad):
            ql, cl = queries.lengths(), candidates.lengths()
            report[f"run{i}_first"] = [(int(q), int(c), int(ql[q]), int(cl[c]), int(run[q, c]) - int(reference[q, c])) for q, c in bad[:6]]
    print(json.dumps(report), flush=True)


lev = lambda: szs.LevenshteinDistances(capabilities=gpu)
nw = lambda: szs.NeedlemanWunschScores(*matrices.blosum62(), open=-4, extend=-4, capabilities=gpu)
rng = np.random.default_rng(11)
tape = lambda count, lo, hi, alphabet: workloads.random_tape(rng, count, lo, hi, alphabet)
for count in (4, 16, 16, 16):
    both(lev, tape(count, 3072, 5120, DNA), tape(count, 3072, 5120, DNA), f"lev {count}x{count} ragged 3072-5120", repeats=4)
both(nw, tape(16, 3072, 5120, PROTEIN), tape(16, 3072, 5120, PROTEIN), "nw 16x16 ragged 3072-5120")
both(nw, tape(64, 600, 1100, PROTEIN), tape(64, 600, 1100, PROTEIN), "nw 64x64 ragged 600-1100")
both(lev, tape(64, 300, 500, DNA), tape(64, 300, 500, DNA), "lev 64x64 single band")
sw = lambda: szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu)
both(sw, tape(16, 3072, 5120, DNA), tape(16, 3072, 5120, DNA), "sw affine 16x16 ragged 3072-5120")

# ---- walk the golden cases in auto mode, announcing each call: which one never returns?
os.environ.pop("SZS_ROCM_TIER", None), os.environ.pop("SZS_ROCM_SWAP", None)
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_matrices.json")
cases = json.load(open(here))
tables = {k: (np.array(v["byte_to_class"], np.uint8), np.array(v["class_costs"], np.int8).reshape(32, 32)) for k, v in cases["tables"].items()}
started = time.perf_counter()
for index, case in enumerate(cases["cases"]):
    queries, candidates = [bytes.fromhex(x) for x in case["queries"]], [bytes.fromhex(x) for x in case["candidates"]]
    if case["kind"] == "levenshtein":
        m, x, o, e = case["costs"]
        engine, dtype = szs.LevenshteinDistances(match=m, mismatch=x, open=o, extend=e, capabilities=gpu), np.uint64
    else:
        cls = szs.NeedlemanWunschScores if case["kind"] == "needleman_wunsch" else szs.SmithWatermanScores
        engine, dtype = cls(*tables[case["table"]], open=case["gaps"][0], extend=case["gaps"][1], capabilities=gpu), np.int64
    print("case", index, case["kind"], case["name"], end=" ", flush=True)
    got = engine(queries, candidates, device=gpu)
    profile = engine.last_call_profile()
    ok = np.array_equal(got, np.array(case["matrix"], dtype=dtype).reshape(len(queries), len(candidates)))
    print("cross", int(profile.tier), int(profile.transposed), ok, end=" ", flush=True)
    sym = engine(queries, device=gpu)
    ok = np.array_equal(sym, np.array(case["symmetric"], dtype=dtype).reshape(len(queries), len(queries)))
    print("sym", int(engine.last_call_profile().tier), ok, round(time.perf_counter() - started, 2), flush=True)
