"""`-m gpu`, round 4: the ONE persistent launch for unit-cost byte calls of mixed lengths (hip/myers_queue.hip) against the oracle.

The reference runs every size tier of a call behind one trampoline (cuda.cuh:4435-4741); rounds 1-3 here launched one kernel
per bit-vector width.  These tests pin: a mixed-width call IS one launch (`profile.launches == 1`, two with the strip kernel's
queries beyond 2048 bytes); every body of the kernel - one lane per pair at 1 ... 8, 10, 12 and 16 words, teams of 2 ... 16
lanes at 4 / 8 / 12 / 16 words per lane - at the edges of its width; candidates per work item from one wave block to several
rounds of the workgroup; symmetric and swapped layouts; both planners; a stream of calls on one engine (the ticket counter is
never reset); and the per-width launches (`queue` knob 0) still agreeing with it cell for cell.
"""
import contextlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import _abi, matrices, workloads  # noqa: E402


@contextlib.contextmanager
def knob(name, value):
    previous = _abi.tuning_set(name, value)
    try:
        yield
    finally:
        _abi.tuning_set(name, previous)


def _rand(rng, count, lo, hi, alphabet):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(count)]


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return szs.DeviceScope(gpu_device=0)


EDGES = [0, 1, 31, 32, 33, 64, 65, 96, 97, 128, 129, 160, 161, 192, 224, 225, 256, 257, 288, 320, 321, 352, 384, 385, 416, 448, 512, 513,
         640, 641, 768, 769, 1024, 1025, 1536, 1537, 2047, 2048]


def test_a_mixed_width_call_is_one_launch(gpu, oracle):
    rng = random.Random(404)
    queries = [bytes(rng.choice(b"ACGT") for _ in range(n)) for n in EDGES]
    candidates = _rand(rng, 70, 0, 300, b"ACGT") + [queries[9], queries[-1][:1900], b""]
    engine = szs.LevenshteinDistances(capabilities=gpu)
    expected = oracle.levenshtein(queries, candidates)
    # 38 x 73 pairs: left alone, the planner would hand this batch to the chained tier, or put the 73 candidates on the workgroups
    with knob("tier", "lanes"), knob("swap", 0):
        got = engine(queries, candidates, device=gpu)
        profile = engine.last_call_profile()
        assert np.array_equal(got, expected), np.argwhere(got != expected)[:5].tolist()
        assert profile.launches == 1 and profile.queue_items > 0 and profile.queue_tiles > 0, (profile.launches, profile.queue_items)
        assert profile.cells == sum(map(len, queries)) * sum(map(len, candidates))
        # queries beyond 2048 bytes keep the strip kernel: one more launch, not one per width
        longer = queries + [bytes(rng.choice(b"ACGT") for _ in range(n)) for n in (2049, 2600)]
        assert np.array_equal(engine(longer, candidates, device=gpu), oracle.levenshtein(longer, candidates))
        assert engine.last_call_profile().launches == 2
        # the per-width launches score the same cells
        with knob("queue", 0):
            assert np.array_equal(engine(queries, candidates, device=gpu), expected)
            assert engine.last_call_profile().launches == 9 and engine.last_call_profile().queue_items == 0
        # a batch of one length class that straddles two widths is not skewed: it keeps its two launches
        flat = _rand(rng, 40, 230, 300, b"ACGT")
        assert np.array_equal(engine(flat, candidates, device=gpu), oracle.levenshtein(flat, candidates))
        assert engine.last_call_profile().queue_items == 0 and engine.last_call_profile().launches == 2
        # a call of ONE width is not queued by itself, but can be
        narrow = [q for q in queries if len(q) <= 256]
        assert np.array_equal(engine(narrow, candidates, device=gpu), oracle.levenshtein(narrow, candidates))
        assert engine.last_call_profile().queue_items == 0
        with knob("queue", 1):
            assert np.array_equal(engine(narrow, candidates, device=gpu), oracle.levenshtein(narrow, candidates))
            assert engine.last_call_profile().queue_items > 0 and engine.last_call_profile().launches == 1


@pytest.mark.parametrize("words", [4, 8, 12, 16])
def test_every_body_of_the_queue_kernel(gpu, oracle, words):
    """`queue_words` caps the words one lane holds: at 4 a 2048-byte query is a team of sixteen lanes and every lane count
    from 2 to 16 occurs (9 ... 64 words), at 16 the one-lane bodies of 10, 12 and 16 words run."""
    rng = random.Random(words)
    lengths = sorted(set(EDGES + [32 * k for k in range(1, 65)] + [32 * k + 1 for k in range(0, 64)] + [rng.randint(1, 2048) for _ in range(20)]))
    queries = [bytes(rng.choice(b"AB") for _ in range(n)) for n in lengths]
    candidates = _rand(rng, 90, 0, 260, b"AB") + _rand(rng, 6, 600, 700, b"AB") + [b"", queries[40]]
    engine = szs.LevenshteinDistances(capabilities=gpu)
    expected = oracle.levenshtein(queries, candidates)
    # (lengths spread evenly up to 2048 are not "skewed": left alone this call keeps its per-width launches, see decide())
    with knob("tier", "lanes"), knob("queue", 1), knob("queue_words", words):
        got = engine(queries, candidates, device=gpu)
        assert engine.last_call_profile().launches == 1 and engine.last_call_profile().queue_items > 0
        wrong = np.argwhere(got != expected)
        assert wrong.size == 0, (words, [(len(queries[q]), len(candidates[c]), int(got[q, c]), int(expected[q, c])) for q, c in wrong[:6]])
        with knob("swap", 1):  # the candidates on the workgroups, the results transposed back
            assert np.array_equal(engine(candidates, queries, device=gpu), expected.T)
        self_expected = oracle.levenshtein(queries[::3], None)
        got = engine(queries[::3], device=gpu)  # symmetric: the lower triangle, mirrored
        assert np.array_equal(got, self_expected)


@pytest.mark.parametrize("rounds", [1, 2, 5])
def test_work_items_of_several_rounds_and_ragged_columns(gpu, oracle, rounds):
    """More candidates than one round of a workgroup: the wavefronts draw wave blocks from the item's counter; columns and
    items that end on no boundary at all; bytes >= 0x80; both planners."""
    rng = random.Random(rounds)
    queries = _rand(rng, 9, 0, 40, bytes(range(256))) + _rand(rng, 5, 257, 700, bytes(range(256))) + _rand(rng, 3, 1025, 1300, bytes(range(256)))
    candidates = _rand(rng, 1337, 0, 48, bytes(range(256))) + _rand(rng, 50, 100, 130, bytes(range(256)))
    engine = szs.LevenshteinDistances(capabilities=gpu)
    expected = oracle.levenshtein(queries, candidates)
    for planner in ("device", "host"):
        with knob("tier", "lanes"), knob("swap", 0), knob("queue_rounds", rounds), knob("planner", planner):
            got = engine(szs.Strs(queries), szs.Strs(candidates), device=gpu)
            profile = engine.last_call_profile()
            assert profile.launches == 1 and profile.queue_items > 0 and profile.planner == (1 if planner == "device" else 0), (profile.launches, profile.planner)
            wrong = np.argwhere(got != expected)
            assert wrong.size == 0, (rounds, planner, wrong[:5].tolist())


def test_a_stream_of_calls_on_one_engine(gpu, oracle):
    """The ticket counter lives in device memory and is never reset: every launch starts where the host knows the last one
    ended.  Batches of other shapes, queued and not, in between."""
    rng = random.Random(9)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    with knob("tier", "lanes"):
        for step in range(12):
            if step % 4 == 3:
                queries, candidates = _rand(rng, 20, 90, 160, b"ACGT"), _rand(rng, 300, 90, 160, b"ACGT")  # one width: the short kernel
            else:
                queries = _rand(rng, rng.randint(3, 40), 0, 200, b"ACGT") + _rand(rng, rng.randint(1, 6), 300, 2048, b"ACGT")
                candidates = _rand(rng, rng.randint(1, 700), 0, rng.choice([20, 300]), b"ACGT")
            with knob("queue", None if step % 2 else 1):  # automatic (skewed batches only) and forced, in turns
                got = engine(queries, candidates, device=gpu)
            assert np.array_equal(got, oracle.levenshtein(queries, candidates)), step
            profile = engine.last_call_profile()
            assert profile.launches == 1 or not profile.queue_items, (step, profile.launches, profile.queue_items)


def test_config5_scaled_both_ways(gpu, oracle):
    """Config 5 (Zipf lengths 8 ... 2048) at 1 / 12 of its side: the queue against the per-width launches and the oracle."""
    load = workloads.config(5, scale=1 / 12)
    queries = [load.queries[i] for i in range(len(load.queries))]
    candidates = [load.candidates[i] for i in range(len(load.candidates))]
    expected = oracle.levenshtein(queries, candidates)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    with knob("tier", "lanes"):
        got = engine(load.queries, load.candidates, device=gpu)
        assert engine.last_call_profile().launches == 1
        assert np.array_equal(got, expected)
        with knob("queue", 0):
            assert np.array_equal(engine(load.queries, load.candidates, device=gpu), expected)
            assert engine.last_call_profile().launches > 1


def _text(rng, count, lo, hi, pools):
    """UTF-8 strings of lo ... hi codepoints drawn from `pools` (lists of characters), as bytes."""
    alphabet = [c for pool in pools for c in pool]
    return ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))).encode() for _ in range(count)]


ASCII = [chr(c) for c in range(0x20, 0x7F)]
CYRILLIC = [chr(c) for c in range(0x410, 0x450)]
LATIN = [chr(c) for c in range(0xC0, 0x17F)]
HAN = [chr(c) for c in range(0x4E00, 0x4E00 + 600)]
HAN_MANY = [chr(c) for c in range(0x4E00, 0x4E00 + 3000)]
EMOJI = [chr(c) for c in range(0x1F600, 0x1F640)]


@pytest.mark.parametrize("words", [None, 4, 8, 12, 16])
@pytest.mark.parametrize("pools", ["small", "medium"])
def test_codepoint_calls_through_the_queue(gpu, oracle, pools, words):
    """The codepoint engine over a batch the device renumbered: ONE scoring launch, tables of alphabet + 1 rows - rows of masks
    for short queries and small alphabets, pointers + a pool of the non-zero chunks for long queries over a rich one."""
    rng = random.Random(len(pools) * 100 + (words or 0))
    chosen = [ASCII, CYRILLIC] if pools == "small" else [ASCII, CYRILLIC, LATIN, HAN, EMOJI]
    queries = _text(rng, 12, 0, 60, chosen) + _text(rng, 6, 250, 700, chosen) + _text(rng, 4, 1000, 2048, chosen) + [b"", "я".encode()]
    queries += ["".join(rng.choice(ASCII + CYRILLIC) for _ in range(n)).encode() for n in (255, 256, 257, 384, 385, 512, 513, 2047, 2048)]
    candidates = _text(rng, 150, 0, 90, chosen) + _text(rng, 12, 300, 800, chosen) + [b"", queries[20]]
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    expected = oracle.levenshtein_utf8(queries, candidates)
    with knob("tier", "lanes"), knob("swap", 0), knob("alphabet", 1), knob("queue", 1), knob("queue_words", words):
        got = engine(queries, candidates, device=gpu)
        profile = engine.last_call_profile()
        wrong = np.argwhere(got != expected)
        assert wrong.size == 0, (pools, words, [(len(queries[q].decode()), len(candidates[c].decode()), int(got[q, c]), int(expected[q, c])) for q, c in wrong[:6]])
        assert profile.launches == 1 and profile.queue_items > 0, (profile.launches, profile.queue_items)
        self_expected = oracle.levenshtein_utf8(queries[::2], None)
        assert np.array_equal(engine(queries[::2], device=gpu), self_expected)
        with knob("swap", 1):
            assert np.array_equal(engine(candidates, queries, device=gpu), expected.T)
    with knob("tier", "lanes"), knob("swap", 0), knob("alphabet", 1), knob("queue", 0):  # the per-width codepoint launches agree
        assert np.array_equal(engine(queries, candidates, device=gpu), expected)
        assert engine.last_call_profile().queue_items == 0


def test_an_alphabet_too_rich_for_the_tables_keeps_the_per_width_launches(gpu, oracle):
    """3,000 distinct runes and a 2,000-rune query: 16 lanes x (3001 x 2 bytes of pointers) + a 32 KB pool is more LDS than a
    workgroup has - the plan says so, the queue stays empty, the per-width kernels (three-level tables) score the call."""
    rng = random.Random(3000)
    queries = _text(rng, 8, 0, 80, [ASCII, HAN_MANY]) + _text(rng, 3, 1900, 2048, [HAN_MANY])
    candidates = _text(rng, 90, 0, 200, [ASCII, HAN_MANY]) + _text(rng, 3, 600, 2048, [HAN_MANY])
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    with knob("tier", "lanes"), knob("swap", 0), knob("alphabet", 1), knob("queue", 1):
        got = engine(queries, candidates, device=gpu)
        assert np.array_equal(got, oracle.levenshtein_utf8(queries, candidates))
        assert engine.last_call_profile().queue_items == 0 and engine.last_call_profile().launches > 1


def test_config5_codepoints_scaled(gpu, oracle):
    """Config 5 at the codepoint level, 1 / 12 of its side, left to itself (renumbering is automatic from the second call of
    a stream on: the engine goes by the cells of its previous call) and with the renumbering pinned."""
    load = workloads.config(6, scale=1 / 12)
    queries = [load.queries[i] for i in range(len(load.queries))]
    candidates = [load.candidates[i] for i in range(len(load.candidates))]
    expected = oracle.levenshtein_utf8(queries, candidates)
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    with knob("tier", "lanes"):
        for _ in range(2):
            assert np.array_equal(engine(load.queries, load.candidates, device=gpu), expected)
        with knob("alphabet", 1):
            assert np.array_equal(engine(load.queries, load.candidates, device=gpu), expected)
            profile = engine.last_call_profile()
            assert profile.launches == 1 and profile.queue_items > 0, (profile.launches, profile.queue_items)


# ---- several GPUs: symmetric calls shard the lower triangle; eight "GPUs" on whatever this box has ------------------------------


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_node_symmetric_calls_score_the_triangle_once(gpu, oracle, devices):
    """`szs_rocm_node_*` with candidates omitted: contiguous bands of rows of equal weight, each a rectangle plus a triangle of its
    own, mirrored after every band has landed - the SAME matrix and the SAME cells as the single-GPU symmetric call (round 3
    scored the full square: twice the work).  The same GPU named eight times rehearses the eight threads, bands and replicas."""
    import torch

    rng = random.Random(len(devices))
    strings = _rand(rng, 150, 0, 300, b"ACGT") + _rand(rng, 6, 600, 2100, b"ACGT") + [b""]
    rng.shuffle(strings)
    node = szs.Node(devices)
    single = szs.LevenshteinDistances(capabilities=gpu)
    expected = single(strings, device=gpu)
    triangle_cells = int(single.last_call_profile().cells)
    assert np.array_equal(expected, oracle.levenshtein(strings, None))
    engine = node.levenshtein_distances()
    assert np.array_equal(engine(strings), expected)  # host tape, host matrix: mirrored on the host
    stats = engine.last_stats
    assert stats["symmetric"] and stats["gpus"] == len(devices) and sum(stats["rows"]) == len(strings)
    assert sum(stats["cells"]) == triangle_cells, (sum(stats["cells"]), triangle_cells)  # the triangle, not the square
    if len(devices) > 1:
        assert max(stats["row_weights"]) <= 1.3 * (sum(stats["row_weights"]) / len(devices))
    tape = szs.Strs(strings).to_device(0)
    out = torch.full((len(strings), len(strings) + 3), -7, dtype=torch.int64, device="cuda")
    engine(tape, out=out[:, :len(strings)])  # device tape, padded device matrix: mirrored by hip/mirror.hip
    assert np.array_equal(out[:, :len(strings)].cpu().numpy().view(np.uint64), expected) and bool((out[:, len(strings):] == -7).all())
    nuc = matrices.nuc44()
    dna = [s for s in strings if len(s) < 700]
    scores = node.smith_waterman_scores(*nuc, open=-4, extend=-1)
    assert np.array_equal(scores(dna).view(np.int64), oracle.smith_waterman(dna, None, *nuc, -4, -1).view(np.int64))
    # fewer rows than GPUs: some bands are empty
    few = strings[:5]
    assert np.array_equal(engine(few), oracle.levenshtein(few, None))
    assert sum(engine.last_stats["rows"]) == 5 and (len(devices) < 8 or 0 in engine.last_stats["rows"])
    # and the ordinary (two-sided) call over the same node, rows dealt by LPT, fewer rows than GPUs included
    others = _rand(rng, 70, 0, 200, b"ACGT")
    assert np.array_equal(engine(strings, others), oracle.levenshtein(strings, others))
    assert np.array_equal(engine(few[:3], others), oracle.levenshtein(few[:3], others))
    assert "peer_pairs" in engine.last_stats and engine.last_stats["peer_pairs"] == 0  # one physical GPU: no pair to enable


# ---- one engine handle through every size and tier (the reference: test/similarities.cuh:2125-2219), and the team tier's limits ---


def test_engines_are_reused_across_sizes_and_tiers(gpu, oracle):
    """ONE handle per family, lengths 128 ... 8192 and back down, batch shapes alternating, ten rounds: the workspaces grow and are
    re-used, the tier changes under the same engine (team <-> packed <-> 32-bit cells, lanes <-> chained, 64-bit cells once by
    knob), and every matrix is checked against the oracle.  The reference sweeps its engines the same way."""
    rng = random.Random(2125)
    nuc, blosum = matrices.nuc44(), matrices.blosum62()
    engines = [
        ("needleman_wunsch", szs.NeedlemanWunschScores(*blosum, open=-4, extend=-4, capabilities=gpu), b"ARNDCQEGHILKMFPSTWYV", (blosum, -4, -4)),
        ("smith_waterman", szs.SmithWatermanScores(*nuc, open=-4, extend=-1, capabilities=gpu), b"ACGT", (nuc, -4, -1)),
        ("levenshtein", szs.LevenshteinDistances(1, 3, 3, 3, capabilities=gpu), b"ACGT", None),
        ("levenshtein", szs.LevenshteinDistances(capabilities=gpu), b"ACGT", None),
    ]
    sizes = [128, 256, 512, 1024, 2048, 8192, 4096, 1024, 128, 8192, 128]
    seen = set()
    for round_index, size in enumerate(sizes):
        wide_batch = round_index % 2 == 0 or size > 1024  # alternate: a few long pairs / many candidates (the oracle scores them all)
        for kind, engine, alphabet, extra in engines:
            queries = _rand(rng, 2 if wide_batch else 5, size // 2, size, alphabet) + [b""]
            candidates = _rand(rng, 3 if wide_batch else 70, size // 3, size, alphabet)
            if kind == "levenshtein":
                costs = (1, 3, 3, 3) if engine is engines[2][1] else (0, 1, 1, 1)
                expected = oracle.levenshtein(queries, candidates, *costs)
            else:
                expected = getattr(oracle, kind)(queries, candidates, *extra[0], extra[1], extra[2])
            with knob("cells", 64 if round_index == 5 and kind != "levenshtein" else None):
                got = engine(queries, candidates, device=gpu)
            profile = engine.last_call_profile()
            seen.add((kind, int(profile.tier), int(profile.cell_bits), int(profile.team) // 10000))
            assert np.array_equal(got.view(np.int64), expected.view(np.int64)), (round_index, size, kind, profile.tier, profile.cell_bits, profile.team)
    tiers = {(kind, bits) for kind, _, bits, _ in seen}
    assert {("needleman_wunsch", 16), ("needleman_wunsch", 32), ("needleman_wunsch", 64), ("smith_waterman", 16), ("levenshtein", 0)} <= tiers, sorted(seen)
    assert any(tier != 0 for _, tier, _, _ in seen), "a few long pairs take a chained tier"


@pytest.mark.parametrize("case", ["global_narrow", "global_wide", "local_narrow", "local_wide", "distance_narrow", "distance_wide"])
def test_team_tier_limits_are_straddled(gpu, oracle, case):
    """Each of the six `team_reach_limit` values (hip/team_core.hpp: 15000 / 29000 / 30000 for cells kept as half-float patterns,
    32000 / 62000 / 64000 as unsigned integers) with the caller's bound one cost step BELOW it and exactly AT it: the host must
    pick the narrow order, the wide one or the 32-bit kernels accordingly, and every score must match the oracle - the only
    guard against a value leaving the range in which `v_pk_maximum3_f16` orders patterns like integers is this bound."""
    rng = random.Random(len(case))
    nuc = matrices.nuc44()  # largest magnitude 5
    objective, wide = case.split("_")
    limit = {"global": (15000, 32000), "local": (29000, 62000), "distance": (30000, 64000)}[objective][wide == "wide"]
    shape = _abi.team_shapes()[0]
    for at_limit in (False, True):
        if objective == "global":  # reach = (rows + columns + 3) x 5, affine gaps
            total = limit // 5 - 3 - (0 if at_limit else 1)
            queries = [bytes(rng.choice(b"ACGT") for _ in range(total // 2))]
            candidates = [bytes(rng.choice(b"ACGT") for _ in range(total - total // 2)), queries[0][: total - total // 2]]
            engine = szs.NeedlemanWunschScores(*nuc, open=-5, extend=-1, capabilities=gpu)
            expected = oracle.needleman_wunsch(queries, candidates, *nuc, -5, -1)
        elif objective == "local":  # bound = (shorter side + 3) x 5
            shorter = limit // 5 - 3 - (0 if at_limit else 1)
            core = bytes(rng.choice(b"ACGT") for _ in range(shorter + 40))
            queries = [core[:shorter]]  # matches its own candidate end to end: the score reaches 5 x shorter, the top of the range
            candidates = [core, bytes(rng.choice(b"ACGT") for _ in range(shorter + 7))]
            engine = szs.SmithWatermanScores(*nuc, open=-5, extend=-1, capabilities=gpu)
            expected = oracle.smith_waterman(queries, candidates, *nuc, -5, -1)
        else:  # reach = (longer side + 1) x 6, costs 2 / 6 / 6 (narrow) or (longer + 3) x 16, costs 0 / 8 / 16 / 4 (wide): strings the
            # device planner takes (under 6 KB) - the batch's byte alphabet is counted on the device, host-planned calls keep 32-bit cells
            costs, per_step, border = ((2, 6, 6, 6), 6, 1) if wide == "narrow" else ((0, 8, 16, 4), 16, 3)
            longer = limit // per_step - border - (0 if at_limit else 1)
            queries = [bytes(rng.choice(b"ACGT") for _ in range(longer))]
            candidates = [bytes(rng.choice(b"ACGT") for _ in range(longer - 5)), b"", queries[0][:100]]
            engine = szs.LevenshteinDistances(*costs, capabilities=gpu)
            expected = oracle.levenshtein(queries, candidates, *costs)
        with knob("team", shape), knob("tier", "lanes"), knob("swap", 0):
            got = engine(queries, candidates, device=gpu)
            profile = engine.last_call_profile()
        assert np.array_equal(got.view(np.int64), expected.view(np.int64)), (case, at_limit, profile.team, profile.team_wide, profile.cell_bits)
        if not at_limit:  # below the limit: this order of cells
            assert profile.team == shape and profile.team_wide == (wide == "wide"), (case, profile.team, profile.team_wide)
        elif wide == "narrow":  # at the narrow limit: the unsigned order takes over
            assert profile.team == shape and profile.team_wide == 1, (case, profile.team, profile.team_wide)
        else:  # at the wide limit: 32-bit cells
            assert profile.team == 0 and profile.cell_bits == 32, (case, profile.team, profile.cell_bits)


def test_transcoding_of_long_lines_well_formed_and_not(gpu, oracle):
    """The transcoder takes well-formed chunks with two ballots and anything else through its chain of jumps (hip/utf8.hip):
    long lines whose sequences straddle the 64-byte chunks at every phase, with and without damage - stray continuation
    bytes, tails cut short, leads at the very end - must decode as `sz_rune_decode_unchecked` does, chunk after chunk."""
    rng = np.random.default_rng(404)
    alphabet = ["a", "b", " ", "é", "ß", "中", "文", "𝄞", "😀"]
    lines = []
    for k in range(24):
        text = "".join(rng.choice(alphabet, size=int(rng.integers(60, 400))))
        raw = bytearray(("x" * (k % 5) + text).encode())
        if k % 3 == 1:  # damage: bytes flipped into strays / leads, one tail cut
            for at in rng.integers(0, len(raw), size=6):
                raw[at] = int(rng.choice([0x80, 0xBF, 0xC3, 0xE4, 0xF0, 0x41]))
        if k % 3 == 2:
            raw = raw[:len(raw) - int(rng.integers(1, 3))] + bytes([0xE4])  # a lead as the last byte
        lines.append(bytes(raw))
    lines += [b"", "é".encode() * 32, ("a" + "中" * 21).encode(), ("ab" + "𝄞" * 16).encode()]
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    expected = oracle.levenshtein_utf8(lines, lines)
    assert np.array_equal(engine(lines, lines, device=gpu), expected)
    with knob("alphabet", 1):
        assert np.array_equal(engine(lines, lines, device=gpu), expected)
    with knob("alphabet", 0):
        assert np.array_equal(engine(lines, lines[:7], device=gpu), oracle.levenshtein_utf8(lines, lines[:7]))
    with knob("planner", "host"), knob("alphabet", 1):  # the host-planned path transcodes refs, and empties the table with fills
        assert np.array_equal(engine(lines, lines, device=gpu), expected)


def test_a_host_planned_call_never_scores_with_the_previous_batch_alphabet(gpu, oracle):
    """Non-unit Levenshtein engines key the team tier's profile by the dense alphabet of THE BATCH (counted on the device).  A
    host-planned call - the `planner` knob, callback sequences, host-only pointers - does not count one: it must not inherit the
    previous call's (bytes absent from the earlier batch all mapped to class 0 and scored as equal)."""
    rng = random.Random(77)
    engine = szs.LevenshteinDistances(0, 2, 3, 3, capabilities=gpu)
    first_q, first_c = _rand(rng, 12, 90, 300, b"ACGT"), _rand(rng, 140, 60, 300, b"ACGT")
    other_q, other_c = _rand(rng, 12, 90, 300, b"klmnopqrstuvwxyz"), _rand(rng, 140, 60, 300, b"klmnopqrstuvwxyz")
    with knob("tier", "lanes"):
        assert np.array_equal(engine(first_q, first_c, device=gpu), oracle.levenshtein(first_q, first_c, 0, 2, 3, 3))
        assert engine.last_call_profile().team, "the device-planned call is expected on the team tier"
        with knob("planner", "host"):
            assert np.array_equal(engine(other_q, other_c, device=gpu), oracle.levenshtein(other_q, other_c, 0, 2, 3, 3))
            assert np.array_equal(engine(other_q, first_c, device=gpu), oracle.levenshtein(other_q, first_c, 0, 2, 3, 3))
        assert np.array_equal(engine(other_q, other_c, device=gpu), oracle.levenshtein(other_q, other_c, 0, 2, 3, 3))


def test_a_stream_of_mixed_length_batches_on_one_engine(gpu, oracle):
    """Batches of one shape after another through the ONE persistent launch, planner free: other strings of the same lengths,
    queries grown to their width group's bound (the slices of the previous plan would not hold them), another shape altogether."""
    rng = random.Random(31)
    load = workloads.config(5, scale=1 / 10)
    queries = [load.queries[i] for i in range(len(load.queries))]
    candidates = [load.candidates[i] for i in range(len(load.candidates))]
    engine = szs.LevenshteinDistances(capabilities=gpu)

    def scored(q, c):
        got = engine(q, c, device=gpu)
        profile = engine.last_call_profile()
        assert np.array_equal(got, oracle.levenshtein(q, c)), (profile.planner, profile.launches)
        return profile

    first = scored(queries, candidates)
    assert first.launches == 1 and first.queue_items > 0, "the batch is expected in the one-launch kernel (skewed lengths)"
    again = scored([q[::-1] for q in queries], [c[::-1] for c in candidates])
    assert again.launches == 1 and again.queue_items == first.queue_items
    bounds = [256, 320, 384, 512, 640, 768, 1024, 1536, 2048]
    longest = max(map(len, queries))
    grown = [(q + bytes(rng.choice(b"abcdef") for _ in range(min(b for b in bounds if b >= len(q)) - len(q))))[:longest] if len(q) > 40 and k % 3 == 0 else q
             for k, q in enumerate(queries)]
    others = [c[: max(1, len(c) - rng.randint(0, 5))] for c in candidates]
    assert scored(grown, others).launches == 1
    assert scored(queries[::2], candidates[5:]).launches == 1


def test_input_formats_and_result_placement_through_the_queue(gpu, oracle):
    """The one-launch kernel behind every way a caller hands strings over and takes results back: 64-bit tapes, plain host
    results (staged copy), a padded device matrix (stride > columns, padding never written: cuda.cuh:2201-2203), a Python list
    (packed into a tape), bytes codepoint engine on the same strings."""
    import torch

    rng = random.Random(64)
    queries = _rand(rng, 30, 0, 120, b"ACGT") + _rand(rng, 5, 300, 2048, b"ACGT")
    candidates = _rand(rng, 333, 0, 140, b"ACGT") + _rand(rng, 7, 500, 900, b"ACGT")
    expected = oracle.levenshtein(queries, candidates)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    with knob("tier", "lanes"), knob("swap", 0):
        wide_q, wide_c = szs.Strs(queries, wide_offsets=True), szs.Strs(candidates, wide_offsets=True)
        assert np.array_equal(engine(wide_q, wide_c, device=gpu), expected)                       # *_u64tape
        assert engine.last_call_profile().queue_items > 0 and engine.last_call_profile().launches == 1
        out = np.full((len(queries), len(candidates)), 0xDEADBEEF, dtype=np.uint64)              # plain host results
        assert engine(queries, candidates, device=gpu, out=out) is out and np.array_equal(out, expected)
        assert engine.last_call_profile().queue_items > 0
        padded = torch.full((len(queries), len(candidates) + 20), -7, dtype=torch.int64, device="cuda")
        view = padded[:, :len(candidates)]
        engine(queries, candidates, device=gpu, out=view)
        assert np.array_equal(view.cpu().numpy().view(np.uint64), expected) and (padded[:, len(candidates):] == -7).all()
        assert engine.last_call_profile().queue_items > 0
        # ASCII through the codepoint engine: scored by the byte kernels (serial.hpp:2809-2813), the same one launch
        utf8 = szs.LevenshteinDistancesUTF8(capabilities=gpu)
        assert np.array_equal(utf8(queries, candidates, device=gpu), expected)
