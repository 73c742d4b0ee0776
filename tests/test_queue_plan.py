"""`-m "not gpu"`: the work queue of the one-launch bit-parallel kernel (hip/myers_queue.hip) as host/plan.c plans it.

`szs_rocm_queue_probe` runs the host planner and `szs_plan_queue` on bare length arrays; this file walks the tickets exactly
the way the kernel does (ticket -> tile -> query, block of candidates cut from the column's end -> wave blocks) and checks
that every (query, candidate) cell of the cross-product is scored exactly once, that every tile's shape takes the queries of
its slice, that the tiles come longest first, and that the shapes follow the size of the call (whole-device batches keep wide
lanes, a short call spreads its pairs)."""
import ctypes

import numpy as np
import pytest

from stringzilla_amd import _abi


def plan(query_lengths, candidate_lengths, symmetric=False, alphabet=0):
    queries = np.ascontiguousarray(query_lengths, dtype=np.uint32)
    candidates = np.ascontiguousarray(candidate_lengths, dtype=np.uint32)
    tiles = np.zeros((96, 10), dtype=np.uint32)
    count, items = ctypes.c_size_t(), ctypes.c_uint64()
    status = _abi.lib.szs_rocm_queue_probe(int(symmetric), alphabet, queries.ctypes.data, len(queries), candidates.ctypes.data, len(candidates),
                                           tiles.ctypes.data, 96, ctypes.byref(count), ctypes.byref(items))
    assert status == 0
    assert count.value <= 96
    return tiles[: count.value].astype(np.int64), int(items.value)


def words_of(length):
    return max(1, -(-int(length) // 32))


def walk(tiles, items, query_lengths, candidate_lengths, alphabet=0):
    """The kernel's own arithmetic.  Returns the coverage matrix [query position (longest first)][candidate position (ascending)]."""
    sorted_queries = np.sort(np.asarray(query_lengths))[::-1]
    covered = np.zeros((len(query_lengths), len(candidate_lengths)), dtype=np.int32)
    expected_first = 0
    rows, arena_dwords = (alphabet + 1, 72 * 256) if alphabet else (256, 64 * 256)
    for first_item, query_first, query_count, c_first, c_end, per_item, words_per_lane, lanes, per_group, flags in tiles:
        assert first_item == expected_first and query_count > 0 and c_end > c_first and per_item > 0 and 1 <= per_group <= 16
        blocks, groups = -(-(c_end - c_first) // per_item), -(-query_count // per_group)
        expected_first += groups * blocks
        longest = int(sorted_queries[query_first])  # slices are cut from the descending array: its first query is its longest
        assert lanes >= 1 and lanes <= 16
        slot_words = (64 // per_group) & ~3  # words of pattern each query of a group may hold
        slot_dwords = (arena_dwords // per_group) & ~3  # dwords of LDS each query's table gets
        needed = words_of(longest)
        if lanes == 1:
            assert needed <= (16 if alphabet else 20), (longest, "one lane per pair takes up to 20 words (16 of codepoints)")
            body = needed if needed <= 8 else (12 if needed <= 12 else 16) if alphabet else 10 if needed <= 10 else 12 if needed <= 12 else 16 if needed <= 16 else 20
            table = rows * ((body + 3) & ~3) if alphabet or body >= 3 else rows * body
            assert body <= slot_words and not flags
        else:
            assert words_per_lane in (4, 8, 12, 16) and needed <= words_per_lane * lanes <= slot_words, (longest, words_per_lane, lanes, per_group)
            if flags & 1:  # pointers (16 bits per chunk of a lane, padded to one read) + a pool of the non-zero chunks
                assert alphabet
                table = -(-(rows * lanes * (2 if words_per_lane <= 4 else 4 if words_per_lane <= 8 else 8)) // 16) * 4 + (longest + 1) * 4
            else:
                table = rows * words_per_lane * lanes
        assert table <= slot_dwords, (longest, lanes, words_per_lane, per_group, flags, table, slot_dwords)
        for local in range(groups * blocks):
            block, group = divmod(local, groups)
            q_first = query_first + group * per_group
            q_count = min(per_group, query_first + query_count - q_first)
            c_hi = c_end - block * per_item
            c_lo = c_hi - per_item if c_hi - c_first > per_item else c_first
            pairs_per_wave = 64 // lanes
            wave_blocks = -(-(c_hi - c_lo) // pairs_per_wave) * q_count
            for drawn in range(wave_blocks):
                candidate_block, g = divmod(drawn, q_count)
                hi = c_hi - candidate_block * pairs_per_wave
                lo = hi - pairs_per_wave if hi - c_lo > pairs_per_wave else c_lo
                covered[q_first + g, lo:hi] += 1
    assert expected_first == items
    return covered


def zipf_lengths(rng, count, low=8, high=2048, exponent=1.1):
    ranks = np.arange(low, high + 1, dtype=np.float64)
    weights = ranks ** (-exponent)
    return rng.choice(np.arange(low, high + 1), size=count, p=weights / weights.sum())


@pytest.mark.parametrize("shape", ["config5", "eighth", "uniform", "few_candidates", "one_each", "ragged", "symmetric"])
def test_every_cell_is_scored_exactly_once(shape):
    rng = np.random.default_rng(len(shape))
    if shape == "config5":
        queries, candidates = zipf_lengths(rng, 3163), zipf_lengths(rng, 3163)
    elif shape == "eighth":
        queries, candidates = zipf_lengths(rng, 395), zipf_lengths(rng, 3163)
    elif shape == "uniform":
        queries, candidates = rng.integers(0, 700, 300), rng.integers(0, 300, 1000)
    elif shape == "few_candidates":
        queries, candidates = rng.integers(1, 2049, 500), rng.integers(0, 2049, 7)
    elif shape == "one_each":
        queries, candidates = np.array([2048]), np.array([5])
    elif shape == "ragged":
        queries = np.concatenate([np.zeros(3, dtype=np.int64), rng.integers(1, 40, 50), [2048, 2047, 1025, 513, 512, 511, 257, 256, 33, 32]])
        candidates = np.concatenate([np.zeros(2, dtype=np.int64), rng.integers(1, 2049, 321)])
    else:
        queries = candidates = zipf_lengths(rng, 700)
    tiles, items = plan(queries, candidates, symmetric=shape == "symmetric")
    covered = walk(tiles, items, queries, candidates)
    assert covered.min() == 1 and covered.max() == 1, (shape, np.argwhere(covered != 1)[:5].tolist())


def test_tiles_come_longest_first_and_shapes_follow_the_call():
    rng = np.random.default_rng(5)
    queries, candidates = zipf_lengths(rng, 3163), zipf_lengths(rng, 3163)
    whole, _ = plan(queries, candidates)
    eighth, _ = plan(np.sort(queries)[::-1][::8], candidates)  # every eighth query, longest first: one GPU's share of eight
    ascending = np.sort(candidates)

    def key(tile):
        _, query_first, _, c_first, c_end, per_item, words_per_lane, lanes, per_group, _ = tile
        pairs_per_wave = 64 // lanes
        wave_blocks = per_group * -(-per_item // pairs_per_wave)
        return -(-wave_blocks // 8) * (words_per_lane if lanes > 1 else 1) * int(ascending[c_end - 1])

    for tiles in (whole, eighth):
        team_tiles = [tile for tile in tiles if tile[7] > 1]
        assert team_tiles, "2048-byte queries are spread over lanes"
        # (one-lane tiles are keyed by their slice's bound, which the probe does not return: check the team tiles' order)
        keys = [key(tile) for tile in team_tiles if tile[4] - tile[3] > 0]
        assert all(earlier * 1.6 >= later for earlier, later in zip(keys, keys[1:])), keys  # sampled lengths: nearly sorted
    # the shape follows the size of the call AND the column: against its longest candidates a short call spreads a pair over
    # more lanes of fewer words; against short candidates it keeps the wide lanes (a third fewer instructions)
    top = lambda tiles: [tile for tile in tiles if tile[7] > 1 and tile[4] == len(candidates)]
    low = lambda tiles: [tile for tile in tiles if tile[7] > 1 and int(ascending[tile[4] - 1]) < 100]
    assert max(tile[6] for tile in top(whole)) >= 12 and max(tile[6] for tile in top(eighth)) <= 4
    assert max(tile[7] for tile in top(eighth)) > max(tile[7] for tile in top(whole))
    assert low(eighth) and min(tile[6] for tile in low(eighth)) >= 8 and max(tile[6] for tile in low(eighth)) >= 12


def test_knobs_pin_the_shape():
    rng = np.random.default_rng(7)
    queries, candidates = zipf_lengths(rng, 500), zipf_lengths(rng, 900)
    for words in (4, 8, 12, 16):
        previous = _abi.tuning_set("queue_words", words)
        try:
            tiles, items = plan(queries, candidates)
        finally:
            _abi.tuning_set("queue_words", previous)
        assert all(tile[6] <= words for tile in tiles if tile[7] > 1)
        covered = walk(tiles, items, queries, candidates)
        assert covered.min() == 1 and covered.max() == 1
    previous = _abi.tuning_set("queue_rounds", 3)
    try:
        tiles, items = plan(queries, candidates)
    finally:
        _abi.tuning_set("queue_rounds", previous)
    for tile in tiles:
        pairs_per_wave = 64 // tile[7]
        assert tile[8] <= 24 and tile[5] == min(max(1, 24 // tile[8]) * pairs_per_wave, tile[4] - tile[3]), tile.tolist()


@pytest.mark.parametrize("alphabet", [60, 255, 926, 1500, 4095])
def test_codepoint_batches_get_tables_that_fit(alphabet):
    """Codepoints of a renumbered batch: tables have alphabet + 1 rows; what does not fit as rows (a 2048-rune query at 926
    symbols would be 237 KB) is pointers + a pool of the non-zero chunks, on teams only; an alphabet too rich for even that
    leaves the queue empty and the call to the per-width launches."""
    rng = np.random.default_rng(alphabet)
    queries, candidates = zipf_lengths(rng, 700), zipf_lengths(rng, 900)
    tiles, items = plan(queries, candidates, alphabet=alphabet)
    longest = int(queries.max())
    pool_fits = lambda lanes, words: -(-((alphabet + 1) * lanes * (2 if words <= 4 else 4 if words <= 8 else 8)) // 16) * 16 + (longest + 1) * 16 <= 72 << 10
    if not items:  # no team shape of the longest query has a table
        assert not any(pool_fits(-(-words_of(longest) // words), words) for words in (4, 8, 12, 16) if -(-words_of(longest) // words) <= 16)
        return
    covered = walk(tiles, items, queries, candidates, alphabet=alphabet)
    assert covered.min() == 1 and covered.max() == 1
    if alphabet >= 926:
        assert any(tile[9] & 1 for tile in tiles), "long codepoint queries over a rich alphabet take the sparse tables"
    if alphabet <= 255:
        assert not any(tile[9] & 1 for tile in tiles), "a small alphabet's tables are rows, like the byte tables"


@pytest.mark.parametrize("seed", range(24))
def test_random_batches_are_covered_exactly_once(seed):
    """Two dozen random batches - counts from 1 to a few thousand, lengths uniform, Zipf, two-peaked or constant, bytes and
    codepoint alphabets small and rich: whatever the plan, every cell once, every table within its share of the arena."""
    rng = np.random.default_rng(1000 + seed)

    def side(count):
        kind = rng.integers(0, 5)
        if kind == 0:
            return rng.integers(0, int(rng.integers(2, 2049)) + 1, count)
        if kind == 1:
            return zipf_lengths(rng, count, low=int(rng.integers(1, 64)), high=int(rng.integers(128, 2049)), exponent=float(rng.uniform(0.8, 1.6)))
        if kind == 2:
            return np.where(rng.random(count) < 0.9, rng.integers(0, 60, count), rng.integers(900, 2049, count))
        if kind == 3:
            return np.full(count, int(rng.integers(0, 2049)))
        return np.concatenate([rng.integers(0, 2049, max(count - 4, 1)), [0, 1, 2047, 2048]])[:max(count, 1)]

    queries, candidates = side(int(rng.integers(1, 900))), side(int(rng.integers(1, 2500)))
    alphabet = int(rng.choice([0, 0, 40, 300, 1500]))
    if alphabet:  # the tables of a rich alphabet hold fewer words: the host only queues what fits (dispatch.c), as here
        queries = np.minimum(queries, 2048 if alphabet <= 300 else 1024)
    tiles, items = plan(queries, candidates, alphabet=alphabet)
    if not len(tiles):
        assert alphabet, "a byte call always has a plan"
        return
    covered = walk(tiles, items, queries, candidates, alphabet=alphabet)
    assert covered.min() == 1 and covered.max() == 1, (seed, alphabet, np.argwhere(covered != 1)[:5].tolist())
