"""The TEAM tier's arithmetic and data flow against the oracle, on the CPU.

`tests/native/team_model.cpp` compiles the kernel's own step / seed / border / profile code (csrc/hip/team_core.hpp) with
g++ and drives it lane by lane the way `weighted_team_kernel` does.  Every engine family, several (lanes, registers)
shapes, ragged candidate blocks, empty sides, asymmetric tables, queries paired with much shorter partners.
"""

import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = os.path.join(ROOT, "tests", "native", "team_model.cpp")
CORE = os.path.join(ROOT, "stringzilla_amd", "csrc", "hip", "team_core.hpp")
LIBRARY = os.path.join(ROOT, "tests", "native", "bin", "libteam_model.so")


@pytest.fixture(scope="module")
def model():
    os.makedirs(os.path.dirname(LIBRARY), exist_ok=True)
    newest = max(os.path.getmtime(SOURCE), os.path.getmtime(CORE))
    if not os.path.exists(LIBRARY) or os.path.getmtime(LIBRARY) < newest:
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-shared", "-fPIC", SOURCE, "-o", LIBRARY], check=True)
    library = ctypes.CDLL(LIBRARY)
    library.team_model_cross.restype = ctypes.c_int
    return library


def run_model(model, local, affine, wide, lanes, registers, queries, candidates, byte_to_class, class_costs, open, extend):
    q_data, q_off = binding.make_tape(queries)
    c_data, c_off = binding.make_tape(candidates)
    results = np.full((len(queries), max(len(candidates), 1)), -777, dtype=np.int64)
    table = np.ascontiguousarray(class_costs, dtype=np.int8).reshape(-1)
    classes = np.ascontiguousarray(byte_to_class, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    status = model.team_model_cross(int(local), int(affine), int(wide), lanes, registers, p(q_data), p(q_off), len(queries), p(c_data),
                                    p(c_off), len(candidates), p(classes), p(table), int(open), int(extend), p(results),
                                    ctypes.c_uint64(results.shape[1]))
    assert status == 0, "shape not instantiated in team_model.cpp"
    return results[:, : len(candidates)]


def random_strings(rng, count, low, high, alphabet):
    letters = np.frombuffer(alphabet, dtype=np.uint8)
    return [letters[rng.integers(0, len(letters), size=int(rng.integers(low, high + 1)))].tobytes() for _ in range(count)]


def random_table(rng, classes=9):
    byte_to_class = np.zeros(256, np.uint8)
    for letter in range(ord("A"), ord("A") + 26):
        byte_to_class[letter] = rng.integers(0, classes)
    table = np.zeros((32, 32), np.int8)
    table[:classes, :classes] = rng.integers(-9, 12, size=(classes, classes))  # asymmetric on purpose (serial.hpp:199-204)
    return byte_to_class, table


SHAPES = [(16, 32), (16, 16), (16, 24), (8, 32), (4, 32), (4, 16), (4, 8), (2, 16), (1, 32), (1, 4), (64, 32), (64, 4), (32, 16)]  # the last three: teams across DPP rows (round 6)


@pytest.mark.parametrize("lanes,registers", SHAPES)
@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("local,affine", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_model_agrees_with_the_oracle(model, lanes, registers, local, affine, wide):
    rng = np.random.default_rng(1000 * lanes + 10 * registers + 2 * local + affine)
    oracle = binding.oracle()
    rows = lanes * registers
    for trial in range(4):
        if trial == 0:
            table = binding.nuc44()
            alphabet = b"ACGTN"
        elif trial == 1:
            table = binding.blosum62()
            alphabet = b"ARNDCQEGHILKMFPSTWYVBZX*"
        else:
            table = random_table(rng)
            alphabet = b"ABCDEFGHIJKLMNOPQRSTUVWXYZab"
        # queries around one, two and a half passes, plus tiny, empty and badly matched partners
        longest = min(2 * rows + rows // 2, 700)
        queries = random_strings(rng, 5, min(max(1, rows - 3), longest // 2), longest, alphabet) + [b"", alphabet[:1], alphabet[:3]]
        queries += random_strings(rng, 2, 1, 12, alphabet)
        if trial == 3:
            queries = queries[:-1]  # an odd count: the last query has no partner
        candidates = random_strings(rng, 300 // lanes + 3, 0, 90, alphabet) + [b""] + random_strings(rng, 3, 1, 5, alphabet)
        open, extend = (int(rng.integers(-7, 0)), int(rng.integers(-3, 1))) if affine else (int(rng.integers(-6, 0)),) * 2
        if trial == 0:
            open, extend = (-4, -1) if affine else (-4, -4)
        got = run_model(model, local, affine, wide, lanes, registers, queries, candidates, *table, open, extend)
        scorer = oracle.smith_waterman if local else oracle.needleman_wunsch
        expected = scorer(queries, candidates, *table, open, extend)
        mismatches = np.argwhere(got != expected)
        assert mismatches.size == 0, (
            f"trial {trial}: first mismatch at {mismatches[0]}: got {got[tuple(mismatches[0])]}, expected "
            f"{expected[tuple(mismatches[0])]} (query {len(queries[mismatches[0][0]])} B, candidate {len(candidates[mismatches[0][1]])} B)"
        )


def test_model_positive_gaps_of_a_global_engine(model):
    """Needleman-Wunsch accepts gap costs of either sign (they are ADDED); the representation must not care."""
    rng = np.random.default_rng(7)
    oracle = binding.oracle()
    table = random_table(rng)
    queries = random_strings(rng, 6, 1, 150, b"ABCDEFGH")
    candidates = random_strings(rng, 40, 0, 60, b"ABCDEFGH")
    for affine, (open, extend) in [(0, (2, 2)), (1, (3, 1)), (1, (-2, 1))]:
        for wide in (0, 1):
            got = run_model(model, 0, affine, wide, 16, 16, queries, candidates, *table, open, extend)
            assert np.array_equal(got, oracle.needleman_wunsch(queries, candidates, *table, open, extend))


@pytest.mark.parametrize("wide", [0, 1])
def test_model_at_the_edge_of_its_range(model, wide):
    """Scores as large as an instance is allowed to see (team_core.hpp: team_reach_limit): long runs of the best-scoring
    symbol, the largest costs a table can hold, the longest gaps.  The host model orders half-float PATTERNS the way the
    hardware orders the floats, so a cell that left the normal range would come out wrong here too."""
    oracle = binding.oracle()
    byte_to_class = np.zeros(256, np.uint8)
    byte_to_class[ord("A")], byte_to_class[ord("B")] = 1, 2
    table = np.zeros((32, 32), np.int8)
    table[1, 1], table[2, 2], table[1, 2], table[2, 1] = 127, 120, -128, -127
    # local: (shorter side + 3) x 127 just below the limit
    limit = 62000 if wide else 29000
    length = limit // 127 - 3
    queries = [b"A" * length, b"A" * (length - 5) + b"B" * 5, b"AB" * (length // 2)]
    candidates = [b"A" * length, b"A" * (length // 2) + b"B" + b"A" * (length // 2 - 1), b"B" * 40]
    for affine, gaps in [(1, (-128, -1)), (0, (-100, -100))]:
        got = run_model(model, 1, affine, wide, 16, 16, queries, candidates, byte_to_class, table, *gaps)
        expected = oracle.smith_waterman(queries, candidates, byte_to_class, table, *gaps)
        assert np.array_equal(got, expected) and expected.max() > 0.9 * limit
    # global: reach (rows + columns + 3) x 128 just below the limit, scores at both ends of the range
    limit = 32000 if wide else 15000
    length = (limit // 128 - 3) // 2
    queries = [b"A" * length, b"B" * length, b"AB" * (length // 2)]
    candidates = [b"A" * length, b"B" * length, b"A", b""]
    for affine, gaps in [(1, (-128, -128 + 1)), (0, (-128, -128)), (1, (-3, -1))]:
        got = run_model(model, 0, affine, wide, 16, 16, queries, candidates, byte_to_class, table, *gaps)
        assert np.array_equal(got, oracle.needleman_wunsch(queries, candidates, byte_to_class, table, *gaps))


@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("lanes,registers", [(16, 32), (4, 32), (16, 16)])
def test_model_scores_weighted_levenshtein(model, lanes, registers, wide):
    """A Levenshtein engine with non-unit costs is the global recurrence over NEGATED costs with the bias at the top of the
    range (team_core.hpp: `distance_`): distances up to 30000 / 64000 in 16 bits.  Uniform costs stand as an identity class
    table here; the kernel builds its profile from class equality instead (GPU tests)."""
    rng = np.random.default_rng(40 + lanes + wide)
    oracle = binding.oracle()
    alphabet = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcde"
    byte_to_class = np.zeros(256, np.uint8)
    for index, letter in enumerate(alphabet):
        byte_to_class[letter] = index + 1
    for match, mismatch, open, extend in [(0, 1, 1, 1), (1, 3, 3, 3), (0, 1, 4, 2), (0, 4, 3, 2), (2, 5, 4, 1)]:
        table = np.full((32, 32), -mismatch, np.int8)
        np.fill_diagonal(table, -match)
        rows = lanes * registers
        queries = random_strings(rng, 5, 1, min(2 * rows, 600), alphabet) + [b"", alphabet[:2]]
        candidates = random_strings(rng, 300 // lanes + 3, 0, 90, alphabet) + [b""]
        got = run_model(model, 2, open != extend, wide, lanes, registers, queries, candidates, byte_to_class, table, -open, -extend)
        expected = oracle.levenshtein(queries, candidates, match, mismatch, open, extend)
        assert np.array_equal(-got, expected.astype(np.int64)), (match, mismatch, open, extend)
    # at the edge of the range: long strings, the largest costs
    limit = 64000 if wide else 30000
    length = limit // 127 - 3
    queries, candidates = [b"A" * length, b"AB" * (length // 2)], [b"B" * length, b"A" * (length - 7), b""]
    for match, mismatch, open, extend in [(0, 127, 127, 127), (3, 127, 127, 126)]:
        table = np.full((32, 32), -mismatch, np.int8)
        np.fill_diagonal(table, -match)
        got = run_model(model, 2, open != extend, wide, 16, 16, queries, candidates, byte_to_class, table, -open, -extend)
        expected = oracle.levenshtein(queries, candidates, match, mismatch, open, extend)
        assert np.array_equal(-got, expected.astype(np.int64)) and expected.max() > 0.9 * limit
