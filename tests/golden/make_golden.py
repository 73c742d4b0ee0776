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


def main_fingerprints():
    """tests/golden/reference_fingerprints.json: (min_hashes, min_counts) the reference's serial fingerprint engines
    produce - the 64-dimension slices (`floating_rolling_hashers`, fingerprints/serial.hpp:1119) and the per-dimension
    fallback (`basic_rolling_hashers`, :646), chosen like its C shim chooses them (c/stringzillas/fingerprints.cuh:49-62)."""
    rng = random.Random(20260923)
    cases = []
    for name, dimensions, widths, seed, shapes in [
        ("defaults_512", 512, None, 0, [(0, 40, b"ACGT"), (90, 130, b"ACGT")]),
        ("one_width_64", 64, [7], 42, [(0, 12, b"AB"), (300, 400, bytes(range(256)))]),
        ("two_widths_128", 128, [4, 9], 7, [(3, 5, b"ACGT"), (4090, 4110, b"ACGT")]),
        ("fallback_100", 100, None, 1, [(0, 70, b"abcdefgh ")]),
        ("fallback_7", 7, [3, 5], 3, [(0, 9, b"xy"), (8190, 8200, b"ACGT")]),
        ("wide_192", 192, [2, 5, 33], 99, [(30, 36, bytes(range(256))), (0, 3, b"z")]),
    ]:
        texts = []
        for lo, hi, alphabet in shapes:
            texts += rand_strings(rng, 4, lo, hi, alphabet)
        hashes, counts, kind = ob.reference_fingerprints(texts, dimensions, widths, seed)
        cases.append({"name": name, "dimensions": dimensions, "window_widths": widths, "seed": seed, "reference_engine": kind,
                      "texts": hexes(texts), "min_hashes": hashes.tolist(), "min_counts": counts.tolist()})
    with open(os.path.join(HERE, "reference_fingerprints.json"), "w") as handle:
        json.dump({"generator": "tests/golden/make_golden.py main_fingerprints()", "cases": cases}, handle)
    print("wrote reference_fingerprints.json:", len(cases), "cases")


def main_utf8():
    """Codepoint-level goldens from the reference's UTF-8 engines (serial.hpp:678,685) -> reference_utf8_matrices.json.
    Alphabets mix 1/2/3/4-byte runes like the reference's own fuzz ("AÉ中😀", test/similarities.cuh:1504); the degenerate
    corpus is test/similarities.py:678-687; lengths are in RUNES."""
    ref = ob.reference(tier=0)
    rng = random.Random(20260922)
    pools = {"mixed": "AÉ中😀", "latin_cyrillic": "abc абв", "cjk": "日本語中文字漢", "ascii": "ABC", "wide": "aé中😀bñ語🚀 "}
    shapes = [("one_by_many", 1, 9, 1, 48), ("many_by_one", 9, 1, 1, 48), ("ragged_square", 6, 6, 0, 40),
              ("rectangular", 4, 40, 1, 48), ("straddles_words", 5, 5, 60, 70), ("beyond_short_kernel", 3, 4, 250, 300)]
    costs_list = [(0, 1, 1, 1), (1, 3, 3, 3), (0, 1, 4, 2), (0, 4, 3, 3)]
    cases = []

    def text(pool, lo, hi):
        return "".join(rng.choice(pool) for _ in range(rng.randint(lo, hi))).encode("utf-8")

    for name, q_count, c_count, lo, hi in shapes:
        for pool_name, pool in pools.items():
            queries = [text(pool, lo, hi) for _ in range(q_count)]
            candidates = [text(pool, lo, hi) for _ in range(c_count)]
            if name == "ragged_square":
                queries[2] = b""
                candidates[4] = b""
            for costs in costs_list:
                cases.append(dict(name=f"{name}/{pool_name}", costs=list(costs), queries=hexes(queries),
                                  candidates=hexes(candidates),
                                  matrix=ref.levenshtein_utf8(queries, candidates, *costs).tolist(),
                                  symmetric=ref.levenshtein_utf8(queries, None, *costs).tolist()))
    degenerate = ["", "x", "é", "🚀", "あ" * 10, "abcXYZ123", "こんにちは世界" * 5, "あ" * 600 + "い" * 20]
    degenerate = [s.encode("utf-8") for s in degenerate]
    for costs in costs_list:
        cases.append(dict(name="degenerate_corpus", costs=list(costs), queries=hexes(degenerate),
                          candidates=hexes(degenerate),
                          matrix=ref.levenshtein_utf8(degenerate, degenerate, *costs).tolist(),
                          symmetric=ref.levenshtein_utf8(degenerate, None, *costs).tolist()))
    out = dict(generator="tests/golden/make_golden.py (main_utf8)",
               source="oracle/_ref/libszs_ref.so = reference v5.1.2 levenshtein_utf8_serial_t / "
                      "affine_levenshtein_utf8_serial_t (serial.hpp:678,685)", cases=cases)
    path = os.path.join(HERE, "reference_utf8_matrices.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__" and "--fingerprints-only" in sys.argv:
    main_fingerprints()
    sys.exit(0)

if __name__ == "__main__":
    if "--utf8-only" not in sys.argv:
        main()
        main_fingerprints()
    main_utf8()
