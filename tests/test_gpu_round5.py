"""`-m gpu`, round 5 against the oracle.

1. The planner INSIDE the scoring launch (hip/lev_myers.hip: `levenshtein_myers_short_fused_kernel`; kernels.h:
   `szs_fused_plan_t`): a unit-cost byte call whose queries fit the short kernel and whose sides hold at most 1024 strings is ONE
   launch whose first two workgroups sort the two sides while the others wait for the refs.  The reference's own fast path
   skips its task pass the same way (cuda.cuh:4297-4340).  Pinned here: such calls report `planner == 4` and `launches == 1`,
   score what the oracle scores over a stream of different batches, ragged shapes, both orientations and both tape flavours;
   a batch that does not fit after all (a query beyond 256 bytes) is planned the ordinary way, never scored wrongly; the plan
   the launch leaves behind serves the same-tapes path; the `fused` knob turns it off.
"""
import contextlib
import json
import os
import ctypes
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import _abi, workloads  # noqa: E402


@contextlib.contextmanager
def knob(name, value):
    previous = _abi.tuning_set(name, value)
    try:
        yield
    finally:
        _abi.tuning_set(name, previous)


def _rand(rng, count, lo, hi, alphabet=b"ACGTN"):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(count)]


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return szs.DeviceScope(gpu_device=0)


@pytest.mark.parametrize("rows,columns,q_span,c_span", [
    (40, 300, (0, 256), (0, 300)),      # ragged both ways, empties, the widest short query
    (257, 513, (90, 170), (1, 90)),     # counts that straddle the blocks of 256; texts shorter than patterns
    (1, 1, (5, 5), (7, 7)),             # a grid of ONE workgroup sorts both sides itself
    (1, 700, (33, 64), (0, 1500)),      # one query; texts beyond the sort's last bin (1023 and more share it)
    (1024, 1024, (8, 40), (8, 40)),     # the most strings a side may hold
    (300, 20, (100, 256), (10, 30)),    # more queries than candidates, long against short: the planner swaps the sides
])
def test_a_short_call_plans_itself_inside_its_launch(gpu, oracle, rows, columns, q_span, c_span):
    rng = random.Random(rows * 1000 + columns)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    modes = []
    for batch in range(4):  # the first call is planned by the planner kernel; every later one of these counts by its own launch
        queries, candidates = _rand(rng, rows, *q_span), _rand(rng, columns, *c_span)
        got = engine(queries, candidates, device=gpu)
        profile = engine.last_call_profile()
        expected = oracle.levenshtein(queries, candidates)
        assert np.array_equal(got, expected), (batch, np.argwhere(got != expected)[:5].tolist())
        assert profile.cells == sum(map(len, queries)) * sum(map(len, candidates))
        assert profile.longest_query == max(map(len, queries)) and profile.longest_candidate == max(map(len, candidates))
        modes.append((profile.planner, profile.launches))
    if modes[0] == (1, 1):  # a first call of one short launch on the lanes tier: the shape the fused launch takes over
        # (3: torch handed the new tapes the previous ones' addresses - tiny batches - and the guarded plan was tried first)
        assert all(mode in ((4, 1), (3, 1)) for mode in modes[1:]) and ((4, 1) in modes or rows * columns < 64), modes
    else:  # another tier or several launches took the first call (a single pair, skewed lengths): no later call may be fused
        assert all(mode[0] != 4 for mode in modes), modes


def test_a_batch_that_no_longer_fits_is_planned_the_ordinary_way(gpu, oracle):
    rng = random.Random(55)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    shape = lambda longest: (_rand(rng, 200, 20, longest), _rand(rng, 260, 0, 200))
    # (no knob is pinned here: a pinned tier or orientation turns speculation, and with it the fused launch, off)
    queries, candidates = shape(256)
    assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
    queries, candidates = shape(256)
    assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
    fused_before = engine.last_call_profile().planner == 4
    # same counts, but one query is now 300 bytes: the sorter blanks its side, the call is planned and scored by two launches
    queries, candidates = shape(256)
    queries[17] = bytes(rng.choice(b"ACGT") for _ in range(300))
    got = engine(queries, candidates, device=gpu)
    expected = oracle.levenshtein(queries, candidates)
    assert np.array_equal(got, expected), np.argwhere(got != expected)[:5].tolist()
    assert engine.last_call_profile().planner != 4
    # ... and the stream goes on: short batches plan themselves again once a short call has been remembered
    for _ in range(3):
        queries, candidates = shape(256)
        assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
    assert engine.last_call_profile().planner == 4 or not fused_before


def test_the_plan_a_fused_launch_leaves_serves_the_same_tapes_and_the_knob_turns_it_off(gpu, oracle):
    rng = random.Random(77)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    first = (szs.Strs(_rand(rng, 300, 64, 200)).to_device(0), szs.Strs(_rand(rng, 500, 32, 220)).to_device(0))
    second = (szs.Strs(_rand(rng, 300, 64, 200)).to_device(0), szs.Strs(_rand(rng, 500, 32, 220)).to_device(0))
    strings = lambda tape: [tape[i] for i in range(len(tape))]
    expected_second = oracle.levenshtein(strings(second[0]), strings(second[1]))
    assert np.array_equal(engine(*first, device=gpu), oracle.levenshtein(strings(first[0]), strings(first[1])))
    assert np.array_equal(engine(*second, device=gpu), expected_second)
    assert engine.last_call_profile().planner == 4
    assert np.array_equal(engine(*second, device=gpu), expected_second)  # the same tapes again: the refs the launch wrote, re-used
    assert engine.last_call_profile().planner == 3
    with knob("fused", 0):
        assert np.array_equal(engine(*first, device=gpu), oracle.levenshtein(strings(first[0]), strings(first[1])))
        assert engine.last_call_profile().planner in (1, 2)


def test_fused_launches_on_64_bit_tapes_and_results_with_a_stride(gpu, oracle):
    import torch

    rng = random.Random(99)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    for batch in range(3):
        queries = szs.Strs(_rand(rng, 130, 0, 256), wide_offsets=True)
        candidates = szs.Strs(_rand(rng, 333, 0, 180), wide_offsets=True)
        out = torch.full((130, 400), -1, dtype=torch.int64, device="cuda:0")
        engine(queries, candidates, device=gpu, out=out[:, :333])
        expected = oracle.levenshtein([queries[i] for i in range(130)], [candidates[i] for i in range(333)])
        assert np.array_equal(out[:, :333].cpu().numpy().view(np.uint64), expected)
        assert (out[:, 333:] == -1).all()  # padding columns are never written (cuda.cuh:2201-2203)
    assert engine.last_call_profile().planner == 4


def test_malformed_offsets_reach_the_caller_from_a_fused_launch_too(gpu):
    """Descending offsets: the sorting workgroup blanks its side and says why; the ordinary planner then reports the error."""
    import torch

    rng = random.Random(5)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    for _ in range(2):
        engine(_rand(rng, 64, 10, 60), _rand(rng, 300, 10, 60), device=gpu)
    queries, candidates = szs.Strs(_rand(rng, 64, 10, 60)).to_device(0), szs.Strs(_rand(rng, 300, 10, 60)).to_device(0)
    _, _, offsets = candidates._device
    offsets[100] = offsets[99] - 5  # string 99 now "ends" before it begins
    with pytest.raises(Exception) as problem:
        engine(queries, candidates, device=gpu)
    assert "ascend" in str(problem.value).lower() or "dimension" in str(problem.value).lower(), str(problem.value)


# ---- 2. the tiny-token kernel (hip/myers_tiny.hip; the reference's fast path: cuda.cuh:2864, :4297-4340) -------------------------------
#
# Straight from the tapes, thirty-two queries per lane on 16-bit bit-vectors, whole runs of the result rows; strings of 17 ... 255
# bytes ride along in the same launch (a block's long candidates under the groups' masks, a span's long queries as W-word patterns).
# Pinned here: tokens at every length 0 ... 16 beside longer ones of up to 255 bytes on either side and on both, ragged counts
# around the blocks of 256 candidates and the groups of 32 queries, bytes >= 0x80, 64-bit tapes, a padded results matrix; a string
# beyond 255 bytes sends the call to the ordinary path; the automatic choice takes word-like batches and nothing else.  The same
# path from plain C on denser mixes (every string long, hundreds of long ones a block): tests/native/words_probe.c, run below.

WORDS = b"etaoinshrdlucmfwypvbgkqjxz" + bytes(range(0xC0, 0xC8)) + b"\xff\x80"


@pytest.mark.parametrize("rows,columns,longest_query,longest_text", [
    (1, 1, 8, 8), (31, 255, 16, 16), (32, 256, 16, 16), (33, 257, 16, 16), (100, 700, 32, 40), (70, 300, 17, 16), (40, 520, 64, 9),
    (65, 260, 128, 70), (20, 270, 256, 300), (300, 1000, 10, 10), (600, 300, 9, 256),
])
def test_tiny_tokens_straight_from_the_tapes(gpu, oracle, rows, columns, longest_query, longest_text):
    rng = random.Random(rows * 7919 + columns)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    previous_fit = False
    with knob("tiny", 1):
        for batch in range(3):
            # mostly word-sized, a few at the longest the shape allows (a group's width follows its longest query), some empty
            length = lambda longest: rng.choice([0, 1, 2, 3, 5, 7, 8, longest, rng.randint(0, longest)])
            queries = [bytes(rng.choice(WORDS) for _ in range(length(longest_query))) for _ in range(rows)]
            candidates = [bytes(rng.choice(WORDS) for _ in range(length(longest_text))) for _ in range(columns)]
            got = engine(queries, candidates, device=gpu)
            expected = oracle.levenshtein(queries, candidates)
            assert np.array_equal(got, expected), (batch, np.argwhere(got != expected)[:5].tolist())
            profile = engine.last_call_profile()
            assert profile.cells == sum(map(len, queries)) * sum(map(len, candidates))
            fits = all(max(map(len, side)) <= 255 for side in (queries, candidates))
            if fits:  # ONE launch: the tiny tokens and, in their shadow, the longer ones (a distance of up to 255 fits a byte of the staging rows)
                assert profile.launches == 1 and profile.planner == (5 if previous_fit else 1), (batch, profile.planner, profile.launches)
            else:  # refused by the kernel (a string beyond 255 bytes): scored by the ordinary path
                assert profile.planner != 5
            previous_fit = fits


@pytest.mark.parametrize("source,rows,columns,padding,wide,knob_value,taken", [
    ("mix:60:40", 100, 700, 0, False, 1, True), ("mix:50:64", 37, 513, 87, True, 1, True), ("mix:20:64", 1, 1, 0, False, 1, True),
    ("mix:300:128", 65, 260, 0, False, 1, False), ("mix:1000:200", 40, 300, 0, False, 1, False), ("mix:500:33", 257, 31, 1, False, 1, False),
    ("mix:100:255", 300, 1000, 7, False, 2, True), ("mix:300:128", 65, 260, 0, False, 2, True), ("mix:1000:200", 40, 300, 0, False, 2, True),
    ("mix:500:33", 257, 31, 1, False, 2, True),
])
def test_dense_mixes_of_tiny_and_long_tokens_from_plain_c(gpu, source, rows, columns, padding, wide, knob_value, taken):
    """tests/native/words_probe.c: device tapes, a padded matrix, every cell against the oracle.  `tiny` = 1: mixes of which more than
    a quarter of a block or span is long are REFUSED by the launch (the ordinary kernels score them - ten times faster there - and
    `planner` says so).  `tiny` = 2 (testing) scores them in the launch all the same: dozens and hundreds of long candidates in a
    block (rounds of sixteen), more long queries in a span than one round of tables holds, W = 1, 2, 4 and 8, strings of exactly
    255 bytes, every string long (the groups' columns idle, kinds A / B / C carry the call)."""
    import subprocess

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "bin", "words_probe")
    if not os.path.exists(probe):
        pytest.skip("tests/native/bin/words_probe is not built (make -C tests/native)")
    env = dict(os.environ, SZS_ROCM_TINY=str(knob_value))
    if wide:
        env["PROBE_WIDE"] = "1"
    done = subprocess.run([probe, source, str(rows), str(columns), "3", str(padding)], env=env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    summary = json.loads(done.stdout.strip().splitlines()[-1])
    assert summary["failures"] == 0 and summary["checked"], summary
    assert (summary["planner"] == 5 and summary["launches"] == 1) == taken, summary


def test_tiny_tokens_are_chosen_for_words_and_for_nothing_else(gpu, oracle):
    rng = random.Random(2024)
    word = lambda: bytes(rng.choice(b"etaoinshrdlu") for _ in range(rng.choice([1, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 11, 14])))
    engine = szs.LevenshteinDistances(capabilities=gpu)
    modes = []
    for batch in range(3):
        queries, candidates = [word() for _ in range(600)], [word() for _ in range(2100)]
        got = engine(queries, candidates, device=gpu)
        assert np.array_equal(got, oracle.levenshtein(queries, candidates))
        modes.append(int(engine.last_call_profile().planner))
    assert modes == [1, 5, 5], modes
    # the same counts, but the strings are sentences now: tried straight from the tapes (the previous call was), refused by the kernel
    # - every string is an outlier - and scored by the ordinary kernels; words after that are recognised again by their summary
    sentence = lambda: b" ".join(word() for _ in range(rng.randint(8, 20)))[:250]
    queries, candidates = [sentence() for _ in range(600)], [sentence() for _ in range(2100)]
    assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
    assert engine.last_call_profile().planner != 5
    modes = []
    for _ in range(2):  # the first may still be scored on the sentences' remembered shape (speculated: 2); its summary says "words"
        queries, candidates = [word() for _ in range(600)], [word() for _ in range(2100)]
        assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
        modes.append(int(engine.last_call_profile().planner))
    assert modes[0] in (1, 2, 4) and modes[1] == 5 and engine.last_call_profile().launches == 1, modes  # (4: round 6, sides beyond 1024 strings plan themselves too)
    # config 2's shape never goes there
    load = workloads.config(2, scale=1 / 4)
    engine(load.queries, load.candidates, device=gpu)
    assert engine.last_call_profile().planner != 5
    with knob("tiny", 0):  # ... and words do not when the knob says so
        queries, candidates = [word() for _ in range(600)], [word() for _ in range(2100)]
        for _ in range(2):
            assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
            assert engine.last_call_profile().planner != 5


def test_a_query_the_tiny_kernel_cannot_hold_sends_the_call_to_the_ordinary_path(gpu, oracle):
    import torch

    rng = random.Random(31)
    word = lambda: bytes(rng.choice(WORDS) for _ in range(rng.randint(0, 12)))
    engine = szs.LevenshteinDistances(capabilities=gpu)
    with knob("tiny", 1):
        for batch in range(2):
            queries, candidates = [word() for _ in range(50)], [word() for _ in range(400)]
            assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
        assert engine.last_call_profile().planner == 5
        queries = [word() for _ in range(50)]
        queries[33] = bytes(rng.choice(WORDS) for _ in range(257))
        candidates = [word() for _ in range(400)]
        got = engine(queries, candidates, device=gpu)  # same counts: tried straight from the tapes, refused by the kernel, planned
        expected = oracle.levenshtein(queries, candidates)
        assert np.array_equal(got, expected), np.argwhere(got != expected)[:5].tolist()
        assert engine.last_call_profile().planner != 5
        # 64-bit tapes, a results matrix with padding columns that must stay untouched
        for batch in range(2):
            queries = szs.Strs([word() for _ in range(37)], wide_offsets=True)
            candidates = szs.Strs([word() for _ in range(513)], wide_offsets=True)
            out = torch.full((37, 600), -1, dtype=torch.int64, device="cuda:0")
            engine(queries, candidates, device=gpu, out=out[:, :513])
            expected = oracle.levenshtein([queries[i] for i in range(37)], [candidates[i] for i in range(513)])
            assert np.array_equal(out[:, :513].cpu().numpy().view(np.uint64), expected)
            assert (out[:, 513:] == -1).all()
        assert engine.last_call_profile().planner == 5


def test_round5_paths_fill_a_host_matrix_too(gpu, oracle):
    """Results in plain host memory (a NumPy `out=`): the library stages a dense matrix in HBM and copies it out with one 2-D copy -
    behind the launch that plans itself and behind the one launch of the tiny-token path alike (cuda.cuh:2205-2215)."""
    rng = random.Random(8)
    # (one token in 25 beyond 16 bytes: 160 of the 4000 candidates, scored by the same launch)
    word = lambda: bytes(rng.choice(b"etaoinshrdlu") for _ in range(rng.choice([0, 1, 2, 3, 3, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 8, 9, 10, 11, 12, 13, 14, 16, 16, 23])))
    engine = szs.LevenshteinDistances(capabilities=gpu)
    modes = []
    for batch in range(3):  # words: summary-chosen, then straight from the tapes
        queries, candidates = [word() for _ in range(300)], [word() for _ in range(4000)]
        out = np.full((300, 4100), 0xEEEE, dtype=np.uint64)
        engine(queries, candidates, device=gpu, out=out[:, :4000])
        assert np.array_equal(out[:, :4000], oracle.levenshtein(queries, candidates)) and (out[:, 4000:] == 0xEEEE).all()
        modes.append(int(engine.last_call_profile().planner))
    assert modes[1:] == [5, 5], modes
    engine = szs.LevenshteinDistances(capabilities=gpu)
    modes = []
    for batch in range(3):  # config 2's shape, an eighth of it: planned, then planning itself
        queries, candidates = _rand(rng, 128, 96, 160), _rand(rng, 1024, 96, 160)
        out = np.full((128, 1024), 0xEEEE, dtype=np.uint64)
        engine(queries, candidates, device=gpu, out=out)
        assert np.array_equal(out, oracle.levenshtein(queries, candidates))
        modes.append(int(engine.last_call_profile().planner))
    assert modes[1:] == [4, 4], modes
