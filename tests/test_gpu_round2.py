"""`-m gpu`: the pieces added in round 2, each against the CPU oracle through the C-ABI, bit-exact.

  - the planner on the device (csrc/hip/planner.hip) and the speculated launches behind it: every family, ragged and
    degenerate shapes, 32- and 64-bit tapes, symmetric calls, batches whose shape changes between calls;
  - the 64-bit cell tier (csrc/hip/wide.hip), forced onto inputs small enough to check (the reference widens its cells
    by the reach rule, serial.hpp:135-162, cuda.cuh:5863-5874);
  - failure paths stay synchronous: a failing call returns with the stream drained and the engine usable.
"""
import contextlib
import ctypes
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import stringzilla_amd as szs  # noqa: E402
from stringzilla_amd import _abi, matrices, workloads  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch

    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    return szs.DeviceScope(gpu_device=0)


@contextlib.contextmanager
def knob(name, value):
    previous = _abi.tuning_set(name, value)
    try:
        yield
    finally:
        _abi.tuning_set(name, previous)  # nested blocks restore the outer setting, not "automatic"


def _strings(rng, count, low, high, alphabet=b"ACGT"):
    return [bytes(rng.choice(alphabet) for _ in range(rng.randint(low, high))) for _ in range(count)]


def _engines(gpu):
    blosum, nuc = matrices.blosum62(), matrices.nuc44()
    return [
        ("lev_unit", szs.LevenshteinDistances(capabilities=gpu), lambda o, q, c: o.levenshtein(q, c)),
        ("lev_weighted", szs.LevenshteinDistances(1, 3, 3, 3, capabilities=gpu), lambda o, q, c: o.levenshtein(q, c, 1, 3, 3, 3)),
        ("lev_affine", szs.LevenshteinDistances(0, 1, 4, 2, capabilities=gpu), lambda o, q, c: o.levenshtein(q, c, 0, 1, 4, 2)),
        ("nw_linear", szs.NeedlemanWunschScores(*blosum, open=-4, extend=-4, capabilities=gpu),
         lambda o, q, c: o.needleman_wunsch(q, c, *blosum, -4, -4)),
        ("nw_affine", szs.NeedlemanWunschScores(*blosum, open=-5, extend=-1, capabilities=gpu),
         lambda o, q, c: o.needleman_wunsch(q, c, *blosum, -5, -1)),
        ("sw_linear", szs.SmithWatermanScores(*nuc, open=-3, extend=-3, capabilities=gpu),
         lambda o, q, c: o.smith_waterman(q, c, *nuc, -3, -3)),
        ("sw_affine", szs.SmithWatermanScores(*nuc, open=-4, extend=-1, capabilities=gpu),
         lambda o, q, c: o.smith_waterman(q, c, *nuc, -4, -1)),
    ]


def test_device_and_host_planners_score_the_same_matrices(gpu, oracle):
    """Every family, a ragged batch: the device-planned call, the same call speculated on the previous shape, and the
    host-planned call all equal the oracle; the profile says which planner ran."""
    rng = random.Random(11)
    queries = _strings(rng, 37, 0, 300, b"ARNDCQEGHILKMFPSTWYV") + [b""]
    candidates = _strings(rng, 301, 0, 200, b"ARNDCQEGHILKMFPSTWYV") + [b"", b"A"]
    for name, engine, expected_of in _engines(gpu):
        expected = expected_of(oracle, queries, candidates).view(np.int64)
        q, c = szs.Strs(queries), szs.Strs(candidates)
        first = engine(q, c, device=gpu).view(np.int64)
        assert engine.last_call_profile().planner == 1, name  # planned on the device, nothing to speculate on yet
        assert np.array_equal(first, expected), name
        second = engine(q, c, device=gpu).view(np.int64)
        profile = engine.last_call_profile()
        # lanes tier: the same tapes again need no planner when the kernels can validate the refs themselves (unit-cost
        # bytes); otherwise the launches go in behind the planner - except non-unit Levenshtein costs over bytes (round 3),
        # whose launch depends on the byte alphabet the device counts for this very call: planned, never speculated
        speculated = 1 if name in ("lev_weighted", "lev_affine") else 2
        # (a re-used plan, 3, and - round 3 - speculation, 2, only where the whole call is ONE launch: this batch's queries of
        # up to 300 bytes are two width groups of the bit-parallel kernels, planned and waited for: 1)
        assert profile.planner in (((1, 2, 3) if name == "lev_unit" else (speculated,)) if profile.tier == 0 else (1,)), name
        assert np.array_equal(second, expected), name
        with knob("reuse", "0"):
            again = engine(q, c, device=gpu).view(np.int64)
            profile = engine.last_call_profile()
            assert profile.planner in (((1, 2) if name == "lev_unit" else (speculated,)) if profile.tier == 0 else (1,)), name
        assert np.array_equal(again, expected), name
        with knob("planner", "host"):
            third = engine(q, c, device=gpu).view(np.int64)
            assert engine.last_call_profile().planner == 0, name
        assert np.array_equal(third, expected), name
        with knob("speculate", "0"):
            fourth = engine(q, c, device=gpu).view(np.int64)
            assert engine.last_call_profile().planner == 1, name
        assert np.array_equal(fourth, expected), name


def test_speculation_survives_a_change_of_shape(gpu, oracle):
    """Same counts, different lengths: launch variants, longest strings and with them the remembered shape change from
    call to call (short -> long queries -> short again -> longer candidates).  Whatever the speculated launches did to the
    results matrix, the call must return the right one."""
    rng = random.Random(5)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    nw = szs.NeedlemanWunschScores(*matrices.blosum62(), open=-4, extend=-4, capabilities=gpu)
    table = matrices.blosum62()
    shapes = [((10, 100), (50, 150)), ((200, 700), (50, 150)), ((10, 100), (50, 150)), ((10, 100), (400, 900)),
              ((0, 40), (0, 30)), ((2100, 2300), (100, 300)), ((10, 100), (50, 150))]
    planners = []
    for (q_low, q_high), (c_low, c_high) in shapes:
        queries = _strings(rng, 24, q_low, q_high, b"ARNDCQEGHILKMFPSTWYV")
        candidates = _strings(rng, 260, c_low, c_high, b"ARNDCQEGHILKMFPSTWYV")
        got = engine(queries, candidates, device=gpu)
        assert np.array_equal(got, oracle.levenshtein(queries, candidates))
        planners.append(engine.last_call_profile().planner)
        got = nw(queries, candidates, device=gpu)
        assert np.array_equal(got, oracle.needleman_wunsch(queries, candidates, *table, -4, -4))
    assert planners[0] == 1 and 1 in planners[1:]  # at least one speculation was refused and re-planned
    # the same batch twice in a row is speculated
    again = engine(queries, candidates, device=gpu)
    assert engine.last_call_profile().planner in (2, 3) and np.array_equal(again, oracle.levenshtein(queries, candidates))


def test_reused_plans_notice_tapes_rewritten_in_place(gpu, oracle):
    """The same device buffers, call after call, with their CONTENTS changing underneath: bytes only (the plan still holds),
    a boundary moved between two strings, every string shortened, the whole tape re-packed.  The kernels validate the refs
    of the previous plan against the offsets as they are now; whatever they find, the call returns the right matrix."""
    import torch

    rng = random.Random(17)
    engine = szs.LevenshteinDistances(capabilities=gpu)

    def pack(strings, capacity):
        offsets = np.zeros(len(strings) + 1, dtype=np.int32)
        np.cumsum([len(s) for s in strings], out=offsets[1:])
        data = np.zeros(capacity, dtype=np.uint8)
        blob = np.frombuffer(b"".join(strings), dtype=np.uint8)
        data[:blob.size] = blob
        return data, offsets

    queries = _strings(rng, 30, 20, 200)
    candidates = _strings(rng, 300, 20, 200)
    capacity_q, capacity_c = 30 * 200 + 64, 300 * 200 + 64
    q_data, q_offsets = (torch.from_numpy(a).cuda() for a in pack(queries, capacity_q))
    c_data, c_offsets = (torch.from_numpy(a).cuda() for a in pack(candidates, capacity_c))
    q_tape = _abi.U32Tape(q_data.data_ptr(), q_offsets.data_ptr(), len(queries))
    c_tape = _abi.U32Tape(c_data.data_ptr(), c_offsets.data_ptr(), len(candidates))
    results = torch.empty((len(queries), len(candidates)), dtype=torch.int64, device="cuda")
    error = ctypes.c_char_p()

    def score_and_check(q_strings, c_strings):
        results.fill_(-1)
        status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(q_tape), ctypes.byref(c_tape),
                                                            results.data_ptr(), len(c_strings), ctypes.byref(error))
        assert status == 0, error.value
        assert np.array_equal(results.cpu().numpy().view(np.uint64), oracle.levenshtein(q_strings, c_strings))
        return engine.last_call_profile().planner

    assert score_and_check(queries, candidates) == 1
    assert score_and_check(queries, candidates) == 3                       # same tapes: the plan is re-used
    # new bytes, same lengths: the plan still describes the tapes
    candidates = [bytes(rng.choice(b"ACGT") for _ in s) for s in candidates]
    c_data.copy_(torch.from_numpy(pack(candidates, capacity_c)[0]).cuda())
    assert score_and_check(queries, candidates) == 3
    # one boundary moves: two candidate refs are stale
    joined = candidates[7] + candidates[8]
    candidates[7], candidates[8] = joined[:5], joined[5:]
    c_offsets.copy_(torch.from_numpy(pack(candidates, capacity_c)[1]).cuda())
    assert score_and_check(queries, candidates) != 3
    assert score_and_check(queries, candidates) == 3
    # every query shortened: every query ref is stale, and would over-read if it were trusted
    queries = [s[: max(1, len(s) // 3)] for s in queries]
    data, offsets = pack(queries, capacity_q)
    q_data.copy_(torch.from_numpy(data).cuda()), q_offsets.copy_(torch.from_numpy(offsets).cuda())
    assert score_and_check(queries, candidates) != 3
    # both tapes re-packed with other strings of other lengths
    queries, candidates = _strings(rng, 30, 0, 190), _strings(rng, 300, 0, 190)
    for tensors, strings, capacity in ((q_data, q_offsets), queries, capacity_q), ((c_data, c_offsets), candidates, capacity_c):
        data, offsets = pack(strings, capacity)
        tensors[0].copy_(torch.from_numpy(data).cuda()), tensors[1].copy_(torch.from_numpy(offsets).cuda())
    assert score_and_check(queries, candidates) != 3
    assert score_and_check(queries, candidates) == 3


def test_device_planner_formats(gpu, oracle):
    """64-bit tapes, symmetric calls, strided result rows, one-string sides and offsets that do not start at zero."""
    import torch

    rng = random.Random(3)
    strings = _strings(rng, 70, 0, 180)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    expected = oracle.levenshtein(strings, strings)
    for wide in (False, True):
        tape = szs.Strs(strings, wide_offsets=wide)
        for _ in range(2):  # plain, then speculated
            assert np.array_equal(engine(tape, device=gpu), expected)  # symmetric
            assert np.array_equal(engine(tape, tape, device=gpu), expected)
    sw = szs.SmithWatermanScores(*matrices.nuc44(), open=-4, extend=-1, capabilities=gpu)
    expected_sw = oracle.smith_waterman(strings, strings, *matrices.nuc44(), -4, -1)
    for _ in range(2):
        assert np.array_equal(sw(szs.Strs(strings), device=gpu), expected_sw)

    # strided results in device memory, padding untouched
    out = torch.full((70, 96), -7, dtype=torch.int64, device="cuda")
    view = out[:, :70]
    engine(szs.Strs(strings), szs.Strs(strings), device=gpu, out=view)
    assert np.array_equal(view.cpu().numpy().view(np.uint64), expected) and bool((out[:, 70:] == -7).all())

    # a tape whose offsets start inside the buffer: sub-tape of a larger one
    whole = szs.Strs(strings)
    whole.to_device(0)
    _, data, offsets = whole._device
    sub = _abi.U32Tape(data.data_ptr(), offsets.data_ptr() + 4 * 10, 50)  # strings 10 .. 59
    results = torch.empty((50, 50), dtype=torch.int64, device="cuda")
    error = ctypes.c_char_p()
    for _ in range(2):
        status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(sub), ctypes.byref(sub),
                                                            results.data_ptr(), 50, ctypes.byref(error))
        assert status == 0, error.value
        assert np.array_equal(results.cpu().numpy().view(np.uint64), expected[10:60, 10:60])

    # 1 x N and N x 1
    one = [strings[5]]
    assert np.array_equal(engine(one, strings, device=gpu), expected[5:6])
    assert np.array_equal(engine(strings, one, device=gpu), expected[:, 5:6])


def test_device_planner_reports_malformed_tapes(gpu):
    import torch

    data = torch.zeros(64, dtype=torch.uint8, device="cuda")
    offsets = torch.tensor([0, 10, 5, 20], dtype=torch.int32, device="cuda")  # descends
    tape = _abi.U32Tape(data.data_ptr(), offsets.data_ptr(), 3)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    results = torch.zeros((3, 3), dtype=torch.int64, device="cuda")
    error = ctypes.c_char_p()
    for _ in range(2):
        status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(tape), ctypes.byref(tape),
                                                            results.data_ptr(), 3, ctypes.byref(error))
        assert status == -15 and b"ascend" in error.value
    good = torch.tensor([0, 10, 15, 20], dtype=torch.int32, device="cuda")
    tape = _abi.U32Tape(data.data_ptr(), good.data_ptr(), 3)
    status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(tape), ctypes.byref(tape),
                                                        results.data_ptr(), 3, ctypes.byref(error))
    assert status == 0  # the engine is still usable
    assert results.cpu().numpy().tolist() == [[0, 5, 5], [5, 0, 0], [5, 0, 0]]


def test_strings_beyond_the_device_planner_fall_back_to_the_host_planner(gpu, oracle):
    rng = random.Random(9)
    queries = _strings(rng, 3, 13000, 14000)  # longer than the planner's histogram
    candidates = _strings(rng, 5, 100, 2000)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    for _ in range(2):
        assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein(queries, candidates))
        assert engine.last_call_profile().planner == 0


@pytest.mark.parametrize("planner", ["host", "device"])
def test_wide_cells_agree_with_the_oracle(gpu, oracle, planner):
    """The 64-bit tier forced onto small inputs: every family, empties, symmetric, both planners."""
    rng = random.Random(21)
    queries = _strings(rng, 9, 0, 90, b"ARNDCQEGHILKMFPSTWYV") + [b""]
    candidates = _strings(rng, 13, 0, 120, b"ARNDCQEGHILKMFPSTWYV") + [b""]
    with knob("cells", "64"), knob("planner", planner):
        for name, engine, expected_of in _engines(gpu):
            got = engine(queries, candidates, device=gpu).view(np.int64)
            assert engine.last_call_profile().cell_bits == 64, name
            assert np.array_equal(got, expected_of(oracle, queries, candidates).view(np.int64)), name
            symmetric = engine(queries, device=gpu).view(np.int64)
            assert np.array_equal(symmetric, expected_of(oracle, queries, queries).view(np.int64)), name
        utf8 = szs.LevenshteinDistancesUTF8(1, 2, 3, 3, capabilities=gpu)
        words = ["naïve", "façade", "日本語のテキスト", "😀 smile", "", "plain ascii"]
        got = utf8(words, device=gpu)
        assert utf8.last_call_profile().cell_bits == 64
        assert np.array_equal(got, oracle.levenshtein_utf8([w.encode() for w in words], [w.encode() for w in words], 1, 2, 3, 3))


def test_wide_cells_with_the_largest_costs(gpu):
    """Costs of magnitude 127 through the 64-bit tier, checked against closed forms: a string against itself scores
    127 n, against the empty string gap x n.  (A pair that really overflows 32 bits needs ~17 M symbols per side - 10^14
    cells - so the tier is exercised through the `cells` knob; the host selects it by the reach rule, serial.hpp:135-162.)"""
    byte_to_class = np.zeros(256, np.uint8)
    costs = np.full((32, 32), -127, np.int8)
    np.fill_diagonal(costs, 127)
    engine = szs.NeedlemanWunschScores(byte_to_class, costs, open=-127, extend=-127, capabilities=gpu)
    text = np.zeros(4096, np.uint8).tobytes()
    with knob("cells", "64"):
        got = engine([text, b""], [text, b""], device=gpu)
    assert engine.last_call_profile().cell_bits == 64
    assert got.tolist() == [[127 * 4096, -127 * 4096], [-127 * 4096, 0]]


def test_failing_calls_leave_the_engine_usable(gpu, oracle):
    """A call that fails after work was enqueued returns with the stream drained (no kernel still writing `results`), and
    the next call on the same engine succeeds."""
    import torch

    strings = [b"kitten", b"sitting", b"saturday", b"sunday"]
    engine = szs.LevenshteinDistances(capabilities=gpu)
    tape = szs.Strs(strings)
    tape.to_device(0)
    q_tape = tape._tape(0)
    results = torch.zeros((4, 4), dtype=torch.int64, device="cuda")
    error = ctypes.c_char_p()
    # stride smaller than the candidate count: refused before anything is enqueued
    status = _abi.lib.szs_levenshtein_distances_u32tape(engine.handle, gpu.handle, ctypes.byref(q_tape), ctypes.byref(q_tape),
                                                        results.data_ptr(), 2, ctypes.byref(error))
    assert status == -15
    assert np.array_equal(engine(strings, strings, device=gpu), oracle.levenshtein(strings, strings))


# ---- the multi-GPU C entry (csrc/host/node.c) on whatever GPUs this box has ---------------------------------------------


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_node_engines_match_the_oracle(oracle, devices):
    """`szs_rocm_node_*`: rows dealt over the node's GPUs (the same GPU named several times exercises the threads, the
    replicas and the row placement on a one-GPU box), every family, tapes in host and in device memory, results into host
    NumPy, device tensors and padded matrices."""
    import torch

    rng = random.Random(77)
    queries = _strings(rng, 41, 0, 260) + [b""]
    candidates = _strings(rng, 67, 0, 300) + [b""]
    node = szs.Node(devices)
    assert len(node) == len(devices)
    nuc = matrices.nuc44()
    cases = [
        (node.levenshtein_distances(), oracle.levenshtein(queries, candidates)),
        (node.levenshtein_distances(1, 3, 4, 2), oracle.levenshtein(queries, candidates, 1, 3, 4, 2)),
        (node.needleman_wunsch_scores(*nuc, open=-4, extend=-1), oracle.needleman_wunsch(queries, candidates, *nuc, -4, -1)),
        (node.smith_waterman_scores(*nuc, open=-3, extend=-3), oracle.smith_waterman(queries, candidates, *nuc, -3, -3)),
    ]
    for engine, expected in cases:
        expected = expected.view(np.int64)
        assert np.array_equal(engine(queries, candidates).view(np.int64), expected)        # host tapes, host matrix
        stats = engine.last_stats
        assert stats["gpus"] == len(devices) and sum(stats["rows"]) == len(queries)
        assert max(stats["row_weights"]) <= 1.25 * (sum(stats["row_weights"]) / len(devices)) + 300  # LPT balance
        q, c = szs.Strs(queries).to_device(0), szs.Strs(candidates).to_device(0)
        out = torch.full((len(queries), len(candidates) + 5), -9, dtype=torch.int64, device="cuda")
        engine(q, c, out=out[:, :len(candidates)])                                         # device tapes, padded device matrix
        assert np.array_equal(out[:, :len(candidates)].cpu().numpy(), expected) and bool((out[:, len(candidates):] == -9).all())
        assert np.array_equal(engine(q, c).view(np.int64), expected)                       # again: replicas and blocks reused
    symmetric = node.levenshtein_distances()(queries)                                      # candidates omitted: the lower triangle, mirrored
    assert np.array_equal(symmetric, oracle.levenshtein(queries, queries))                 # (tests/test_gpu_round4.py has the bands)
    utf8 = node.levenshtein_distances_utf8()
    words = [w.encode() for w in ["naïve", "façade", "日本語", "😀 smile", "", "plain", "naive", "facade"]]
    assert np.array_equal(utf8(words, words), oracle.levenshtein_utf8(words, words))


def test_node_probe_from_plain_c():
    """The same entry driven from C without Python or torch in the process (tests/native/node_probe.c)."""
    import json
    import os
    import subprocess

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "bin", "node_probe")
    if not os.path.exists(probe):
        pytest.skip("tests/native/bin/node_probe is not built")
    for family, arguments in [("lev", ["90", "130", "0", "400"]), ("nw", ["40", "70", "0", "200"]), ("sw", ["33", "300", "5", "150"])]:
        for gpus in (["0"], ["0", "0"], ["0"] * 8):  # eight "GPUs": the world size BASELINE.json's curve ends at
            done = subprocess.run([probe, family, *arguments, *gpus], capture_output=True, text=True, timeout=300)
            assert done.returncode == 0, (family, gpus, done.stdout, done.stderr)
            report = json.loads(done.stdout.strip().splitlines()[-1])
            assert report["mismatches"] == 0 and report["gpus"] == len(gpus)


# ---- 64-bit tapes at scale: offsets beyond 2^32 (reference stringzillas.h:83-92) -------------------------------------------


@pytest.mark.parametrize("planner", ["device", "host"])
def test_u64_tape_with_offsets_beyond_4_gib(gpu, oracle, planner):
    """A `sz_sequence_u64tape_t` over a 4.5 GB buffer whose strings all start beyond byte 2^32 - a corpus that does not fit
    a 32-bit tape.  5 GB of HBM is nothing on a 288 GB part; the strings themselves are small, so the oracle checks every
    cell.  Both planners, every engine family that reads a tape."""
    import torch

    rng = random.Random(13)
    base = (1 << 32) + 12345
    strings = _strings(rng, 40, 0, 300, b"ACGT")
    lengths = np.array([len(s) for s in strings], dtype=np.uint64)
    offsets = np.zeros(len(strings) + 1, dtype=np.uint64)
    offsets[0] = base
    np.cumsum(lengths, out=offsets[1:])
    offsets[1:] += np.uint64(base)
    data = torch.empty(base + int(lengths.sum()) + 64, dtype=torch.uint8, device="cuda")  # contents below 2^32 never read
    payload = torch.from_numpy(np.frombuffer(b"".join(strings), dtype=np.uint8).copy()).cuda()
    data[base:base + payload.numel()] = payload
    device_offsets = torch.from_numpy(offsets.view(np.int64)).cuda()
    tape = _abi.U64Tape(data.data_ptr(), device_offsets.data_ptr(), len(strings))
    results = torch.empty((len(strings), len(strings)), dtype=torch.int64, device="cuda")
    error = ctypes.c_char_p()
    nuc = matrices.nuc44()
    cases = [
        (szs.LevenshteinDistances(capabilities=gpu), _abi.lib.szs_levenshtein_distances_u64tape, oracle.levenshtein(strings, strings)),
        (szs.LevenshteinDistancesUTF8(capabilities=gpu), _abi.lib.szs_levenshtein_distances_utf8_u64tape,
         oracle.levenshtein_utf8(strings, strings)),
        (szs.NeedlemanWunschScores(*nuc, open=-4, extend=-1, capabilities=gpu), _abi.lib.szs_needleman_wunsch_scores_u64tape,
         oracle.needleman_wunsch(strings, strings, *nuc, -4, -1)),
        (szs.SmithWatermanScores(*nuc, open=-4, extend=-1, capabilities=gpu), _abi.lib.szs_smith_waterman_scores_u64tape,
         oracle.smith_waterman(strings, strings, *nuc, -4, -1)),
    ]
    with knob("planner", planner):
        for engine, call, expected in cases:
            for candidates in (tape, None):  # rectangular call, then the symmetric one
                results.fill_(-1)
                status = call(engine.handle, gpu.handle, ctypes.byref(tape), None if candidates is None else ctypes.byref(candidates),
                              results.data_ptr(), len(strings), ctypes.byref(error))
                assert status == 0, error.value
                assert np.array_equal(results.cpu().numpy(), expected.view(np.int64))
    del data
    torch.cuda.empty_cache()


# ---- real text through the reference's benchmark tokeniser (bench/shared.hpp:240-290) ------------------------------------


@pytest.mark.parametrize("tokens", ["words", "lines"])
def test_real_text_tokens_match_the_oracle(gpu, oracle, tokens):
    """The only real prose in this image is the documentation that travels with the repository (a few
    hundred KB of English, Markdown tables and code, with multi-byte punctuation): tokenised the way the reference's bench
    does (`workloads.tokenize_dataset`), 400 x 400 tokens drawn like `bench/similarities.cuh` draws them, scored at the byte
    and at the codepoint level, every cell against the oracle.  Real text is ragged (words of 1 .. 40 bytes, lines of
    0 .. 1100) and repetitive (duplicates, shared prefixes) in ways the synthetic configs are not."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    corpus = b"".join(open(os.path.join(root, name), "rb").read() for name in ("SURVEY.md", "DESIGN.md", os.path.join("docs", "history", "DESIGN_rounds_1_to_4.md")))
    found = workloads.tokenize_dataset(corpus, tokens)
    assert len(found) > 300
    rng = np.random.default_rng(7)
    queries = [found[int(i)] for i in rng.integers(0, len(found), size=400)]
    candidates = [found[int(i)] for i in rng.integers(0, len(found), size=400)]
    assert any(max(q) >= 0x80 for q in queries + candidates if q), "the sample should hold multi-byte text"
    for engine, expected in ((szs.LevenshteinDistances(capabilities=gpu), oracle.levenshtein(queries, candidates)),
                             (szs.LevenshteinDistancesUTF8(capabilities=gpu), oracle.levenshtein_utf8(queries, candidates)),
                             (szs.LevenshteinDistances(0, 2, 3, 1, capabilities=gpu), oracle.levenshtein(queries, candidates, 0, 2, 3, 1))):
        for _ in range(2):
            assert np.array_equal(engine(queries, candidates, device=gpu), expected)


# ---- several lanes per pair for the long byte widths (lev_myers.hip: levenshtein_myers_split_kernel) ----------------------


@pytest.mark.parametrize("lanes", ["2", "4", "8", None])
def test_split_lanes_agree_with_the_oracle(gpu, oracle, lanes):
    """Queries of 385 .. 2048 bytes (launch variants 16, 20 - round 3: 20 words run in the 24-word kernel - 24, 32, 48, 64 words)
    with every pair spread over 2, 4 or 8 lanes: every width boundary, ragged candidates from empty to longer than the queries,
    more candidates than one workgroup takes, symmetric."""
    rng = random.Random(64 + int(lanes or 0))
    lengths = (385, 450, 512, 513, 600, 640, 641, 700, 768, 769, 800, 1023, 1024, 1025, 1100, 1500, 1536, 1537, 1600, 2000, 2047, 2048)
    queries = [bytes(rng.choice(b"ACGTN") for _ in range(n)) for n in lengths]
    candidates = _strings(rng, 150, 0, 2300, b"ACGTN") + [b"", b"A", queries[3], queries[-1][:-1]]
    engine = szs.LevenshteinDistances(capabilities=gpu)
    expected = oracle.levenshtein(queries, candidates)
    with knob("split", lanes), knob("tier", "lanes"), knob("swap", "0"):
        for _ in range(2):  # planned, then the plan re-used (the split kernels carry the guard too)
            assert np.array_equal(engine(queries, candidates, device=gpu), expected)
        assert np.array_equal(engine(queries, device=gpu), oracle.levenshtein(queries, queries))
    with knob("split", "0"), knob("tier", "lanes"), knob("swap", "0"):
        assert np.array_equal(engine(queries, candidates, device=gpu), expected)


@pytest.mark.parametrize("lanes,rune_ids", [("2", None), ("4", None), (None, None), ("4", "40"), ("2", "3")])
def test_split_lanes_of_codepoints_agree_with_the_oracle(gpu, oracle, lanes, rune_ids):
    """The codepoint twin: queries of 385 .. 2048 runes from a 300-rune alphabet of 1 .. 4-byte sequences, pairs over 2 or 4
    lanes; with `rune_ids` the match-mask table is shrunk so that most runes overflow it and travel beside the deltas."""
    rng = random.Random(640 + int(lanes or 0) + int(rune_ids or 0))
    alphabet = [chr(c) for c in list(range(0x41, 0x5B)) + list(range(0x3B1, 0x3C9)) + list(range(0x4E00, 0x4EF0)) + list(range(0x1F600, 0x1F60A))]
    text = lambda n: "".join(rng.choice(alphabet) for _ in range(n)).encode()
    lengths = (385, 512, 513, 640, 641, 768, 769, 1023, 1024, 1025, 1536, 1537, 2000, 2047, 2048)
    queries = [text(n) for n in lengths]
    candidates = [text(rng.randrange(0, 2300)) for _ in range(140)] + [b"", "\u03b1".encode(), queries[3], queries[-1][:-4]]
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    expected = oracle.levenshtein_utf8(queries, candidates)
    with knob("split", lanes), knob("rune_ids", rune_ids), knob("tier", "lanes"), knob("swap", "0"):
        for _ in range(2):
            assert np.array_equal(engine(queries, candidates, device=gpu), expected)
        assert np.array_equal(engine(queries, device=gpu), oracle.levenshtein_utf8(queries, queries))
    with knob("split", "0"), knob("tier", "lanes"), knob("swap", "0"):
        assert np.array_equal(engine(queries, candidates, device=gpu), expected)


@pytest.mark.parametrize("alphabet,rune_ids", [("1", None), ("0", None), ("1", "3"), (None, None)])
def test_renumbered_alphabet_agrees_with_the_oracle(gpu, oracle, alphabet, rune_ids):
    """The codepoint engine with the batch's runes renumbered 1 ... A on the device (hip/utf8.hip; `alphabet` 1) and without
    (0): every kernel family of the rune path - the short mixed-width launch, the long widths one lane per pair and split over
    lanes, the strips beyond 2048 runes - on 1 .. 4-byte sequences, empties, symmetric; with `rune_ids` most runes overflow the
    per-query table.  Then a batch of more distinct runes than the direct tables hold (it must keep its runes and still score)."""
    rng = random.Random(900 + int(alphabet or 7) + int(rune_ids or 0))
    letters = [chr(c) for c in list(range(0x30, 0x7B)) + list(range(0x3B1, 0x3C9)) + list(range(0x4E00, 0x4F40)) + list(range(0x1F600, 0x1F620))]
    text = lambda n, source=letters: "".join(rng.choice(source) for _ in range(n)).encode()
    lengths = [0, 1, 2, 31, 32, 33, 100, 255, 256, 257, 300, 400, 512, 640, 700, 1000, 1024, 1500, 2047, 2048, 2049, 2300]
    queries = [text(n) for n in lengths]
    candidates = [text(rng.randrange(0, 700)) for _ in range(280)] + [text(rng.randrange(1500, 2400)) for _ in range(12)] + [b"", queries[5]]
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    expected = oracle.levenshtein_utf8(queries, candidates)
    with knob("alphabet", alphabet), knob("rune_ids", rune_ids), knob("tier", "lanes"), knob("swap", "0"):
        assert np.array_equal(engine(queries, candidates, device=gpu), expected)
        assert np.array_equal(engine(queries, device=gpu), oracle.levenshtein_utf8(queries, queries))
        with knob("swap", "1"):
            assert np.array_equal(engine(queries, candidates, device=gpu), expected)
    if rune_ids is None:
        wide = [chr(c) for c in range(0x4E00, 0x4E00 + 6000)]  # 6000 distinct runes: beyond SZS_ALPHABET_MOST
        queries = [text(n, wide) for n in (10, 200, 300, 900)] + ["".join(wide).encode()]
        candidates = [text(rng.randrange(0, 500), wide) for _ in range(70)] + ["".join(wide[::-1]).encode()]
        with knob("alphabet", alphabet), knob("tier", "lanes"):
            assert np.array_equal(engine(queries, candidates, device=gpu), oracle.levenshtein_utf8(queries, candidates))


@pytest.mark.parametrize("seed", range(4))
def test_codepoint_fuzz(gpu, oracle, seed):
    """Random shapes through the codepoint engine with the planner free: up to 160 x 300 strings, rune lengths 0 .. 700 (every residue
    of the four-columns-per-load main loop and its tail), alphabets from 3 to 3000 runes of mixed UTF-8 widths, with the batch's
    runes renumbered (`alphabet` 1), kept (0) and left to the size rule."""
    rng = random.Random(4200 + seed)
    pool = [chr(c) for c in rng.sample(list(range(0x21, 0x7F)) + list(range(0xA1, 0x250)) + list(range(0x400, 0x500)) + list(range(0x4E00, 0x5600)) +
                                       list(range(0x1F300, 0x1F400)), rng.choice([3, 40, 300, 3000]))]
    text = lambda n: "".join(rng.choice(pool) for _ in range(n)).encode()
    high = rng.choice([5, 40, 300, 700])
    queries = [text(rng.randint(0, high)) for _ in range(rng.randint(1, 160))]
    candidates = [text(rng.randint(0, high)) for _ in range(rng.randint(1, 300))]
    engine = szs.LevenshteinDistancesUTF8(capabilities=gpu)
    expected = oracle.levenshtein_utf8(queries, candidates)
    for alphabet in ("1", "0", None):
        with knob("alphabet", alphabet):
            assert np.array_equal(engine(queries, candidates, device=gpu), expected), (seed, alphabet, len(queries), len(candidates), high, len(pool))
    with knob("alphabet", "1"):
        assert np.array_equal(engine(queries, device=gpu), oracle.levenshtein_utf8(queries, queries))


@pytest.mark.parametrize("merge", ["1", "3", "8", None])
def test_merged_candidate_blocks_agree_with_the_oracle(gpu, oracle, merge):
    """The short bit-parallel kernels with several candidate blocks per workgroup (`merge` pins the number; None: the launcher's
    rule): 1500 candidates = 6 blocks, the last one ragged, so groups of 3 and of 8 end early; bytes and codepoints, every
    query width of the short launch, empties, symmetric, plan re-use."""
    rng = random.Random(5100 + int(merge or 0))
    queries = [bytes(rng.choice(b"abcdefgh ") for _ in range(n)) for n in [0, 1, 5, 31, 32, 33, 64, 100, 128, 200, 255, 256] + [rng.randint(1, 40) for _ in range(40)]]
    candidates = _strings(rng, 1500, 0, 60, b"abcdefgh ") + [b"", queries[7]]
    lev, utf8 = szs.LevenshteinDistances(capabilities=gpu), szs.LevenshteinDistancesUTF8(capabilities=gpu)
    accented = [q.replace(b"a", "\u00e1".encode()).replace(b"e", "\u20ac".encode()) for q in queries]
    accented_candidates = [c.replace(b"a", "\u00e1".encode()).replace(b"e", "\u20ac".encode()) for c in candidates]
    with knob("merge", merge), knob("tier", "lanes"), knob("swap", "0"):
        expected = oracle.levenshtein(queries, candidates)
        for _ in range(2):
            assert np.array_equal(lev(queries, candidates, device=gpu), expected)
        assert np.array_equal(lev(candidates[:700], device=gpu), oracle.levenshtein(candidates[:700], candidates[:700]))
        for alphabet in ("0", "1"):
            with knob("alphabet", alphabet):
                assert np.array_equal(utf8(accented, accented_candidates, device=gpu), oracle.levenshtein_utf8(accented, accented_candidates))


def test_profile_cells_belong_to_the_batch_that_was_scored(gpu):
    """Two batches of one shape alternate (every call speculated on the other's plan), then one of them repeats (its plan
    re-used): `cells` in the call profile - what bench.py turns into GCUPS - is always the current batch's, never the one the
    remembered decision was first made for."""
    rng = np.random.default_rng(77)
    first = workloads.random_tape(rng, 64, 96, 160, workloads.ASCII_PRINTABLE).to_device(0), workloads.random_tape(rng, 300, 96, 160, workloads.ASCII_PRINTABLE).to_device(0)
    second = workloads.random_tape(rng, 64, 96, 160, workloads.ASCII_PRINTABLE).to_device(0), workloads.random_tape(rng, 300, 96, 160, workloads.ASCII_PRINTABLE).to_device(0)
    cells = lambda pair: int(pair[0].lengths().sum()) * int(pair[1].lengths().sum())
    assert cells(first) != cells(second)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    seen = []
    for fused in ("0", None):  # round 5: such calls plan themselves inside their launch (mode 4) unless the `fused` knob says no (mode 2)
        with knob("fused", fused):
            for pair in (first, second, first, second, second, second, first, first):
                engine(pair[0], pair[1], device=gpu)
                profile = engine.last_call_profile()
                seen.append(int(profile.planner))
                assert int(profile.cells) == cells(pair), (seen, int(profile.cells), cells(first), cells(second))
    assert 2 in seen and 3 in seen and 4 in seen, seen  # the speculated, the re-used and the self-planned path were all taken
