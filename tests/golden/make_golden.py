#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REAL reference engines.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

It drives oracle/_ref/libszs_ref.so - the reference's own serial engines (`levenshtein_serial_t`,
`needleman_wunsch_serial_t`, `smith_waterman_serial_t` and the affine siblings; include/stringzillas/similarities/
serial.hpp:677-689) compiled header-only from /root/reference by oracle/Makefile - on seeded inputs, and stores the
inputs (hex) next to the matrices the reference produced.  The GPU box has no /root/reference, so these committed
vectors are what pins the oracle and the HIP path there.

`known_answers.json` is different: it is hand-transcribed from the reference's own tests (each entry cites its
source line) and is NOT produced by running anything - it is what the reference's authors assert.
"""

from __future__ import annotations

import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rand_strings(rng, count, lo, hi, alphabet):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(count)]


def hexes(strings):
    return [s.hex() for s in strings]


def main():
    ob.build(with_reference=True)
    ref = ob.reference(tier=0)
    rng = random.Random(20260921)
    cases = []

    # Shapes follow the reference's cross-product tests (test/similarities.cuh:1283-1326): 1xN, Nx1, 1x1, ragged
    # square with empties, rectangular, empty sides; fuzz alphabet "ABC" and lengths 1..200 follow
    # test/stringzilla.hpp:395-400; cost schemes follow test/similarities.cuh:722-763.
    shapes = [
        ("one_by_many", 1, 9, 1, 60),
        ("many_by_one", 9, 1, 1, 60),
        ("one_by_one", 1, 1, 20, 200),
        ("ragged_square", 7, 7, 0, 40),
        ("rectangular", 5, 11, 1, 130),
        ("straddles_words", 6, 6, 60, 70),
        ("longer", 3, 4, 150, 300),
    ]
    lev_costs = [(0, 1, 1, 1), (1, 3, 3, 3), (0, 1, 4, 2), (0, 4, 3, 2)]
    for name, q_count, c_count, lo, hi in shapes:
        for alphabet_name, alphabet in (("abc", b"ABC"), ("bytes", bytes(range(256)))):
            queries = rand_strings(rng, q_count, lo, hi, alphabet)
            candidates = rand_strings(rng, c_count, lo, hi, alphabet)
            if name == "ragged_square":
                queries[2] = b""
                candidates[5] = b""
            for costs in lev_costs:
                cases.append(
                    dict(
                        kind="levenshtein", name=f"{name}/{alphabet_name}", costs=list(costs),
                        queries=hexes(queries), candidates=hexes(candidates),
                        matrix=ref.levenshtein(queries, candidates, *costs).tolist(),
                        symmetric=ref.levenshtein(queries, None, *costs).tolist(),
                    )
                )

    tables = {"blosum62": ob.reference_table(0), "nuc44": ob.reference_table(1)}
    asym_rng = random.Random(99)
    asym_map = np.array([asym_rng.randint(0, 31) for _ in range(256)], dtype=np.uint8)
    asym_tab = np.array([[asym_rng.randint(-9, 9) for _ in range(32)] for _ in range(32)], dtype=np.int8)
    tables["asymmetric_random"] = (asym_map, asym_tab)
    gap_schemes = [(-4, -4), (-4, -1), (-1, -1), (-11, -2)]
    for name, q_count, c_count, lo, hi in shapes:
        for table_name, alphabet in (("blosum62", b"ARNDCQEGHILKMFPSTWYV"), ("nuc44", b"ACGT"),
                                     ("asymmetric_random", bytes(range(256)))):
            queries = rand_strings(rng, q_count, lo, hi, alphabet)
            candidates = rand_strings(rng, c_count, lo, hi, alphabet)
            if name == "ragged_square":
                queries[1] = b""
                candidates[0] = b""
            byte_to_class, class_costs = tables[table_name]
            for gaps in gap_schemes:
                for kind in ("needleman_wunsch", "smith_waterman"):
                    fn = getattr(ref, kind)
                    cases.append(
                        dict(
                            kind=kind, name=f"{name}/{table_name}", table=table_name, gaps=list(gaps),
                            queries=hexes(queries), candidates=hexes(candidates),
                            matrix=fn(queries, candidates, byte_to_class, class_costs, *gaps).tolist(),
                            symmetric=fn(queries, None, byte_to_class, class_costs, *gaps).tolist(),
                        )
                    )

    out = dict(
        generator="tests/golden/make_golden.py",
        source="oracle/_ref/libszs_ref.so = reference v5.1.2 serial engines (serial.hpp:677-689), g++ header-only build",
        tables={k: dict(byte_to_class=v[0].tolist(), class_costs=np.asarray(v[1]).reshape(-1).tolist())
                for k, v in tables.items()},
        cases=cases,
    )
    path = os.path.join(HERE, "reference_matrices.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
