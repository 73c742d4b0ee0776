"""`-m gpu`, round 3: the TEAM tier of the 16-bit weighted scorers (hip/weighted_teams.hip) against the oracle.

Every compiled shape (lanes per team x registers per lane) x {Needleman-Wunsch, Smith-Waterman} x {linear, affine} x
{NUC.4.4, BLOSUM62, a 32-class asymmetric table}: query lengths around one, two and two-and-a-half passes, partners of very
different length (the halves of a register), an odd query count, empty strings on both sides, ragged candidate blocks,
symmetric mode and swapped sides.  tests/test_team_model.py checks the same arithmetic on the CPU.
"""
import contextlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import _abi, matrices  # noqa: E402


@contextlib.contextmanager
def forced_env(name, value):
    previous = _abi.tuning_set(name, value)
    try:
        yield
    finally:
        _abi.tuning_set(name, previous)  # nested blocks restore the outer setting, not "automatic"


def forced_tier(name):
    return forced_env("tier", name)


def _rand(rng, count, lo, hi, alphabet):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(count)]


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return szs.DeviceScope(gpu_device=0)


def _tables(rng):
    asym_map = np.array([rng.randint(0, 31) for _ in range(256)], dtype=np.uint8)
    asym_map[:32] = np.arange(32, dtype=np.uint8)  # every class in use
    asym_tab = np.array([[rng.randint(-9, 9) for _ in range(32)] for _ in range(32)], dtype=np.int8)
    return [(matrices.nuc44(), b"ACGTN"), (matrices.blosum62(), b"ARNDCQEGHILKMFPSTWYVBZX*"), ((asym_map, asym_tab), bytes(range(256)))]


def test_team_shapes_are_listed():
    shapes = _abi.team_shapes()
    assert shapes and all(shape // 10000 in (1, 2, 4, 8, 16, 32, 64) and shape // 100 % 100 % 4 == 0 for shape in shapes)


@pytest.mark.parametrize("kind", ["needleman_wunsch", "smith_waterman"])
@pytest.mark.parametrize("shape", _abi.team_shapes())
def test_team_tier_agrees_with_the_oracle(gpu, oracle, kind, shape):
    lanes, registers = shape // 10000, shape // 100 % 100
    rows = lanes * registers
    rng = random.Random(shape * 2 + (kind == "smith_waterman"))
    cls = szs.NeedlemanWunschScores if kind == "needleman_wunsch" else szs.SmithWatermanScores
    for table_index, ((byte_to_class, class_costs), alphabet) in enumerate(_tables(rng)):
        for gaps in [(-4, -4), (-5, -1)] + ([(-2, 0)] if table_index == 0 else []):
            engine = cls(byte_to_class, class_costs, open=gaps[0], extend=gaps[1], capabilities=gpu)
            longest = min(2 * rows + rows // 2, 1100)
            lengths = [0, 1, 2, 3, registers - 1, registers, registers + 1, rows - 1, rows, rows + 1, 2 * rows, longest, longest - 7]
            lengths += [rng.randint(1, longest) for _ in range(6)]
            lengths = [min(length, longest) for length in lengths]  # (the wave-wide shape's 2 x rows would be 4096: past the 16-bit reach)
            queries = [bytes(rng.choice(alphabet) for _ in range(max(0, length))) for length in lengths]  # an odd count
            candidates = _rand(rng, 300 // lanes + 5, 0, 120, alphabet) + _rand(rng, 9, 0, 3, alphabet) + [b""]
            expected = getattr(oracle, kind)(queries, candidates, byte_to_class, class_costs, *gaps)
            with forced_env("team", shape), forced_tier("lanes"):
                got = engine(queries, candidates, device=gpu)
                profile = engine.last_call_profile()
                if lanes > 16 and table_index > 0 and profile.team == 0:
                    continue  # sixty-four strips of a rich alphabet do not fit a CU's LDS: the knob's shape is not taken, nothing to check
                assert profile.team == shape and profile.cell_bits == 16, (profile.team, profile.cell_bits)
                wrong = np.argwhere(got != expected)
                assert wrong.size == 0, (kind, shape, gaps, table_index, wrong[:5].tolist(), got[tuple(wrong[0])], expected[tuple(wrong[0])],
                                         len(queries[wrong[0][0]]), len(candidates[wrong[0][1]]))
                if table_index == 0:
                    expected_sym = getattr(oracle, kind)(queries, None, byte_to_class, class_costs, *gaps)
                    assert np.array_equal(engine(queries, device=gpu), expected_sym), (kind, shape, gaps, "symmetric")
                    with forced_env("SZS_ROCM_SWAP", "1"):
                        assert np.array_equal(engine(queries, candidates, device=gpu), expected), (kind, shape, gaps, "swapped")
                        assert engine.last_call_profile().transposed == 1 and engine.last_call_profile().team == shape
            with forced_env("team", 0), forced_tier("lanes"):  # the knob's other setting: one pair per lane
                assert np.array_equal(engine(queries, candidates, device=gpu), expected), (kind, gaps, "team off")
                assert engine.last_call_profile().team == 0


@pytest.mark.parametrize("kind", ["needleman_wunsch", "smith_waterman"])
def test_team_tier_long_candidates(gpu, oracle, kind):
    """Candidates long enough for the unpredicated main loop to carry most of the columns, lengths ragged inside a wavefront
    and across the workgroup; more than one candidate block; scores far from the bias."""
    rng = random.Random(11)
    cls = szs.NeedlemanWunschScores if kind == "needleman_wunsch" else szs.SmithWatermanScores
    table, alphabet = (matrices.nuc44(), b"ACGT")
    for shape in _abi.team_shapes():
        lanes, registers = shape // 10000, shape // 100 % 100
        for gaps in [(-4, -1), (-3, -3)]:
            engine = cls(*table, open=gaps[0], extend=gaps[1], capabilities=gpu)
            queries = _rand(rng, 5, lanes * registers - 40, lanes * registers + 60, alphabet)
            candidates = _rand(rng, 256 // lanes + 3, 200, 700, alphabet) + _rand(rng, 3, 17, 40, alphabet)
            expected = getattr(oracle, kind)(queries, candidates, *table, *gaps)
            with forced_env("team", shape), forced_tier("lanes"):
                got = engine(queries, candidates, device=gpu)
                assert engine.last_call_profile().team == shape
                wrong = np.argwhere(got != expected)
                assert wrong.size == 0, (kind, shape, gaps, wrong[:5].tolist())


@pytest.mark.parametrize("kind", ["needleman_wunsch", "smith_waterman"])
def test_team_tier_cell_orders(gpu, oracle, kind):
    """The team tier keeps its cells as half-float patterns (three-input maxima) while every DP value provably stays inside
    15 bits, as unsigned integers (two-input maxima) up to 16, and leaves the call to the 32-bit kernels beyond
    (csrc/hip/team_core.hpp).  One batch on each side of both limits, scores near the top of the range included."""
    rng = random.Random(31)
    cls = szs.NeedlemanWunschScores if kind == "needleman_wunsch" else szs.SmithWatermanScores
    table, alphabet = matrices.blosum62(), b"ARNDCQEGHILKMFPSTWYV"  # largest magnitude 11
    shape = _abi.team_shapes()[0]
    # global: reach = (rows + columns + 3) x 11; local: (shorter side + 3) x 11
    sizes = {"needleman_wunsch": [(600, 0, True), (1300, 1, True), (2000, None, False)],
             "smith_waterman": [(2500, 0, True), (3000, 1, True), (6000, None, False)]}[kind]
    for length, wide, teamed in sizes:
        engine = cls(*table, open=-11, extend=-1, capabilities=gpu)
        core = bytes(rng.choice(alphabet) for _ in range(length))
        queries = [core, core[: length - 9] + b"WWWWWWWWW", bytes(rng.choice(alphabet) for _ in range(length - 3))]
        candidates = [core, core[5:] + b"ARNDC"] + _rand(rng, 6, length - 40, length, alphabet) + [b"", b"W"]
        expected = getattr(oracle, kind)(queries, candidates, *table, -11, -1)
        with forced_env("team", shape), forced_tier("lanes"):
            got = engine(queries, candidates, device=gpu)
            profile = engine.last_call_profile()
            assert (profile.team == shape) == teamed and (not teamed or profile.team_wide == wide), (length, profile.team, profile.team_wide)
            assert np.array_equal(got, expected), (kind, length)


def _profile_fits(shape, classes):
    """csrc/hip/team_core.hpp: team_profile_layout - does the cost profile of `classes` classes fit a CU's LDS?"""
    lanes, registers = shape // 10000, shape // 100 % 100
    row_bytes = 4 * registers
    if lanes < 16:
        return classes * (lanes * (row_bytes + 16) + 16) + 16 <= 160 * 1024 - 4096
    slots = 1 if row_bytes >= 256 else min(256 // row_bytes, lanes)
    blocks, class_bytes = lanes // slots, 256 if slots > 1 else row_bytes
    return blocks * (classes * class_bytes + 16) <= 160 * 1024 - 4096


@pytest.mark.parametrize("costs", [(1, 3, 3, 3), (0, 1, 4, 2), (0, 4, 3, 2), (2, 5, 4, 1), (0, 1, 2, 2)])
def test_weighted_levenshtein_on_the_team_tier(gpu, oracle, costs):
    """Non-unit Levenshtein costs over byte tapes: the team tier over the negated costs, its profile keyed by the dense
    alphabet of the batch (counted on the device per call), 16-bit cells - against the oracle, with the knob both ways, for
    alphabets of 4, 95 and 256 byte values (sixteen strips of 256 classes do not fit a CU's LDS: four lanes take those)."""
    rng = random.Random(hash(costs) & 0xFFFF)
    engine = szs.LevenshteinDistances(*costs, capabilities=gpu)
    for alphabet, q_low, q_high, q_count, c_count, c_high in [
        (b"ACGT", 90, 700, 9, 130, 300), (bytes(range(32, 127)), 96, 160, 12, 300, 160), (bytes(range(256)), 100, 1200, 7, 70, 200),
        (b"AB", 1, 60, 9, 40, 50),
    ]:
        queries = _rand(rng, q_count, q_low, q_high, alphabet) + [b"", alphabet[:1]]
        candidates = _rand(rng, c_count, 0, c_high, alphabet) + [b""]
        expected = oracle.levenshtein(queries, candidates, *costs)
        expected_sym = oracle.levenshtein(queries, None, *costs)
        with forced_tier("lanes"):
            got = engine(queries, candidates, device=gpu)
            automatic = engine.last_call_profile()
            assert np.array_equal(got, expected), (costs, len(alphabet), automatic.team)
            assert np.array_equal(engine(queries, device=gpu), expected_sym), (costs, len(alphabet), "symmetric")
            for shape in _abi.team_shapes():
                with forced_env("team", shape):
                    assert np.array_equal(engine(queries, candidates, device=gpu), expected), (costs, len(alphabet), shape)
                    profile = engine.last_call_profile()
                    fits = _profile_fits(shape, len(alphabet))
                    assert (profile.team == shape and profile.cell_bits == 16) if fits else profile.cell_bits == 32, (shape, profile.team, profile.cell_bits)
            with forced_env("team", 0):
                assert np.array_equal(engine(queries, candidates, device=gpu), expected), (costs, len(alphabet), "team off")
                assert engine.last_call_profile().team == 0 and engine.last_call_profile().cell_bits == 32
    # the long, costly end of the 16-bit range and just beyond it (reach = (longest + 1 or 3) x largest cost)
    queries, candidates = _rand(rng, 3, 5000, 6000, b"ACGT"), _rand(rng, 20, 3000, 6000, b"ACGT")
    expected = oracle.levenshtein(queries, candidates, *costs)
    with forced_tier("lanes"), forced_env("team", _abi.team_shapes()[0]):
        assert np.array_equal(engine(queries, candidates, device=gpu), expected), (costs, "long")
        longest = max(len(s) for s in queries + candidates)
        reach = (longest + (1 if costs[2] == costs[3] else 3)) * max(costs)  # serial.hpp:135-162, minimising
        profile = engine.last_call_profile()
        assert profile.team_wide == (1 if reach >= 30000 else 0) or profile.team == 0, (reach, profile.team, profile.team_wide)


@pytest.mark.parametrize("costs", [(0, 1, 1, 1), (1, 3, 3, 3), (0, 1, 4, 2)])
def test_codepoint_engine_planned_on_the_device(gpu, oracle, costs):
    """Round 3 plans codepoint calls over tapes on the device too: both tapes are transcoded without the host reading an
    offset, the planner sorts by RUNE count, one wait (csrc/host/ways_runes.c: szs_cross_device_planned_runes).  Against the oracle
    and against the host-planned path (`planner` knob), for: mixed scripts, an ASCII corpus (byte engines), symmetric calls, a
    batch that outgrows the UTF-32 buffer of the call before, strings longer than the planner's histogram, 64-bit tapes."""
    rng = random.Random(hash(costs) & 0xFFF)
    pools = ["AÉ中😀", "abc абв", "aé中😀bñ語🚀 ", "".join(chr(0x4E00 + i) for i in range(400))]
    engine = szs.LevenshteinDistancesUTF8(*costs, capabilities=gpu)  # one engine: buffers persist from call to call
    for round_, (low, high, q_count, c_count) in enumerate([(0, 48, 7, 300), (1, 20, 64, 5), (200, 300, 3, 70), (900, 1200, 2, 30)]):
        for pool in pools:
            text = lambda: "".join(rng.choice(pool) for _ in range(rng.randint(low, high))).encode()
            queries, candidates = [text() for _ in range(q_count)] + [b""], [text() for _ in range(c_count)] + [b"", "é".encode()]
            expected = oracle.levenshtein_utf8(queries, candidates, *costs)
            got = engine(queries, candidates, device=gpu)
            assert np.array_equal(got, expected), (pool[:4], low, high, "device-planned")
            assert engine.last_call_profile().planner in (1, 2)  # 2: launched on the previous batch's shape (test below)
            with forced_env("planner", "host"):
                assert np.array_equal(engine(queries, candidates, device=gpu), expected), (pool[:4], low, high, "host-planned")
                assert engine.last_call_profile().planner == 0
            if round_ == 0:
                assert np.array_equal(engine(queries, device=gpu), oracle.levenshtein_utf8(queries, None, *costs)), (pool[:4], "symmetric")
    # an ASCII corpus through the codepoint engine: the byte kernels take over
    strings = [bytes(rng.choice(b"ACGT ") for _ in range(rng.randint(0, 90))) for _ in range(40)]
    assert np.array_equal(engine(strings, device=gpu), oracle.levenshtein(strings, None, *costs))
    # beyond the planner's histogram (6143 symbols): the host planner takes over
    long_queries = ["".join(rng.choice("aé中") for _ in range(6500)).encode(), "é".encode() * 300]
    long_candidates = ["".join(rng.choice("aé中") for _ in range(length)).encode() for length in (6400, 100, 0)]
    assert np.array_equal(engine(long_queries, long_candidates, device=gpu), oracle.levenshtein_utf8(long_queries, long_candidates, *costs))
    # a large batch right after small ones: the UTF-32 buffer grows inside the call
    big = ["".join(rng.choice("abcdé語") for _ in range(rng.randint(50, 400))).encode() for _ in range(700)]
    fresh = szs.LevenshteinDistancesUTF8(*costs, capabilities=gpu)
    fresh([b"a"], [b"b"], device=gpu)
    assert np.array_equal(fresh(big[:40], big, device=gpu), oracle.levenshtein_utf8(big[:40], big, *costs))


@pytest.mark.parametrize("renumbered", [None, "1"])
def test_codepoint_batches_of_one_shape_are_scored_without_a_wait(gpu, oracle, renumbered):
    """A stream of codepoint batches of one shape (round 3): tapes transcoded, renumbered, planned AND scored behind one
    another, the host waiting once at the end - `planner` 2 in the call profile.  The planner refuses the speculated launches
    (every ref blank, the call planned afresh, `planner` 1) when the batch has another shape, more runes than the UTF-32
    buffer holds or - with the runes renumbered - more distinct ones than the kernels' direct tables have rows."""
    rng = random.Random(77 + int(renumbered or 0))
    plain = "abcdefghijklmnopqrstuvwxyz é"
    rich = plain + "".join(chr(0x4E00 + i) for i in range(300))

    def batch(pool, high, exact=None):
        text = lambda n: "".join(rng.choice(pool) for _ in range(n)).encode()
        queries = [text(exact or rng.randint(10, high)) for _ in range(40)] + [text(high)]
        candidates = [text(exact or rng.randint(0, high)) for _ in range(280)] + [text(high), b""]
        return queries, candidates

    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    planners = []
    with forced_env("alphabet", renumbered):
        for pool, high in ((plain, 200), (plain, 200), (plain, 180), (plain, 200), (rich, 200), (rich, 190), (plain, 200), (plain, 256), (plain, 256)):
            queries, candidates = batch(pool, high)
            assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates)), (len(planners), planners)
            planners.append(int(engine.last_call_profile().planner))
        # 0: the first call; 4: (renumbered) 300 more distinct runes than the tables were sized for; 7: longer strings than the
        # launches were shaped for
        assert planners == [1, 2, 2, 2, 1 if renumbered else 2, 2, 2, 1, 2], planners
        # an ASCII batch of the remembered shape: renumbered tables refuse it (the byte engines take over), else the codepoint
        # kernels score it - runes that happen to be bytes
        queries, candidates = batch("acgt ", 256)
        assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))
        # the same shape in four-byte sequences: more runes than the buffer of the calls before holds - refused, grown, planned
        small = szs.LevenshteinDistancesUTF8(capabilities=gpu)
        queries, candidates = batch(plain, 230, exact=230)
        assert np.array_equal(small(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))
        queries, candidates = batch("".join(chr(0x1F600 + i) for i in range(40)), 230, exact=230)
        assert np.array_equal(small(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))
        assert small.last_call_profile().planner == 1
        queries, candidates = batch("".join(chr(0x1F600 + i) for i in range(40)), 230, exact=230)
        assert np.array_equal(small(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))
        assert small.last_call_profile().planner == 2
        with forced_env("speculate", "0"):
            assert np.array_equal(small(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))
            assert small.last_call_profile().planner == 1
