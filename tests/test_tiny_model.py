"""`-m "not gpu"`: the arithmetic and the LDS layouts of the tiny-token launch (hip/myers_tiny.hip) restated in Python.

What the kernel does to one workgroup's share of a call - a block of up to 256 candidates against a span of queries - is replayed
here step by step on plain integers, so that a change of a layout can be checked WITHOUT a GPU:

  * a group of 2 R = 32 queries lives two to a 32-bit register: slot s in half s / R of register s % R, its pattern RIGHT-ALIGNED in
    the half's sixteen bits (bit 16 - len + i for byte i; the low rows are phantoms whose deltas start and stay at zero);
  * `peq[byte][R]` holds the group's match masks; one column of the packed recurrence (`tiny_column`) advances both halves at once -
    the addition must not carry from the low half into the high one (`v_pk_add_u16`), nor the shifts (`v_pk_lshlrev_b16`);
  * a distance is the text's length plus popcount(VP) - popcount(VN) of its half; `out[j][column]` packs the slots j, j + H, R + j,
    R + j + H a byte each (H = R / 2), and row s of the group reads byte s / H of `out[s % H]`;
  * kind A: a long candidate (17 ... 255 bytes) is the text of a cluster of R lanes, lane d advancing register d alone; its two
    distances land in bytes d / H and 2 + d / H of `out[d % H][its column]`;
  * kinds B / C: a long query is a W-word pattern (W = 1, 2, 4, 8 by the longest of the span), right-aligned in 32 W bits with the
    carry of the addition rippling through the words (`myers_column`, hip/myers_core.hpp), under tiny and long candidates alike;
  * a block or span of which more than a quarter is long is refused.
Every cell of the replayed workgroup is compared with the plain dynamic programme (the oracle's when it is built)."""
import random

import pytest

R, H, GROUP = 16, 8, 32
MASK32 = 0xFFFFFFFF
LONGEST = 255


def levenshtein(a, b):
    row = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        previous, row[0] = row[0], i
        for j, y in enumerate(b, 1):
            previous, row[j] = row[j], min(row[j] + 1, row[j - 1] + 1, previous + (x != y))
    return row[-1]


def pk_add(a, b):  # v_pk_add_u16
    return (((a & 0xFFFF) + (b & 0xFFFF)) & 0xFFFF) | ((((a >> 16) + (b >> 16)) & 0xFFFF) << 16)


def pk_shl1(a):  # v_pk_lshlrev_b16 by one
    return (((a & 0xFFFF) << 1) & 0xFFFF) | ((((a >> 16) << 1) & 0xFFFF) << 16)


def tiny_column(vp, vn, eq):
    """`tiny_column`: one DP column of two 16-row patterns; returns the new (vp, vn)."""
    xv = eq | vn
    total = pk_add(eq & vp, vp)
    d0 = (total ^ vp) | eq
    hp = (vn | ~(d0 | vp)) & MASK32
    hn = vp & d0
    hp_shifted = pk_shl1(hp) | 0x00010001
    hn_shifted = pk_shl1(hn)
    return (hn_shifted | ~(xv | hp_shifted)) & MASK32, hp_shifted & xv


def start_of(low, high):
    """VP of a register whose halves hold patterns of `low` and `high` rows (None: no tiny pattern in that half)."""
    low_rows, high_rows = low or 0, high or 0
    return ((0xFFFF << (16 - low_rows)) & 0xFFFF) | ((0xFFFF0000 << (16 - high_rows)) & 0xFFFF0000)


def half_distance(length, vp, vn, half):
    return length + bin((vp >> (16 * half)) & 0xFFFF).count("1") - bin((vn >> (16 * half)) & 0xFFFF).count("1")


def myers_column(vp, vn, eq):
    """`myers_column<W>` (hip/myers_core.hpp): one column of a W-word pattern, one carry chain; lists updated in place."""
    carry, hp_below, hn_below = 0, 0, 0
    for w in range(len(vp)):
        xv = eq[w] | vn[w]
        total = (eq[w] & vp[w]) + vp[w] + carry
        carry, total = total >> 32, total & MASK32
        d0 = (total ^ vp[w]) | eq[w]
        hp = (vn[w] | ~(d0 | vp[w])) & MASK32
        hn = vp[w] & d0
        hp_shifted = ((hp << 1) | (1 if w == 0 else hp_below >> 31)) & MASK32
        hn_shifted = ((hn << 1) | (0 if w == 0 else hn_below >> 31)) & MASK32
        hp_below, hn_below = hp, hn
        vp[w] = (hn_shifted | ~(xv | hp_shifted)) & MASK32
        vn[w] = hp_shifted & xv


def long_pattern_distance(pattern, text, words):
    """Kinds B / C: `pattern` (17 ... 255 bytes) right-aligned in 32 `words` bits against `text`."""
    rows = 32 * words
    pad = rows - len(pattern)
    table = {}
    for p, byte in enumerate(pattern):  # the round's table: bit pad + p of the row of the pattern's p-th byte
        bit = pad + p
        table.setdefault(byte, [0] * words)[bit >> 5] |= 1 << (bit & 31)
    vp = [MASK32 if 32 * w >= pad else (0 if 32 * w + 32 <= pad else (MASK32 << (pad - 32 * w)) & MASK32) for w in range(words)]
    vn = [0] * words
    for byte in text:
        myers_column(vp, vn, table.get(byte, [0] * words))
    return len(text) + sum(bin(x).count("1") for x in vp) - sum(bin(x).count("1") for x in vn)


def replay_workgroup(queries, candidates, dense=False):
    """One workgroup's share: every cell of `queries` x `candidates` (at most 256 of them), the way the launch computes it.
    Returns None when the launch refuses the share."""
    assert len(candidates) <= 256
    if any(len(s) > LONGEST for s in queries + candidates):
        return None
    long_columns = [c for c, text in enumerate(candidates) if len(text) > 16]
    long_queries = [q for q, pattern in enumerate(queries) if len(pattern) > 16]
    if not dense and (len(long_columns) > 256 // 4 or len(long_queries) > max(len(queries) // 4, 4)):
        return None
    results = [[None] * len(candidates) for _ in queries]
    for group_first in range(0, len(queries), GROUP):
        group = queries[group_first:group_first + GROUP]
        lengths = [len(q) if len(q) <= 16 else None for q in group] + [None] * (GROUP - len(group))  # None: skipped (long, or past the end)
        peq = {}  # [byte][register]
        for slot, pattern in enumerate(group):
            if lengths[slot] is None:
                continue
            for position, byte in enumerate(pattern):
                peq.setdefault(byte, [0] * R)[slot % R] |= (0x10000 if slot // R else 1) << (16 - len(pattern) + position)
        out = [[0] * 256 for _ in range(H)]  # [j][column], four bytes each
        for column, text in enumerate(candidates):
            if len(text) <= 16:  # the group's columns: a lane, sixteen registers
                packed = [0] * H
                for d in range(R):
                    vp, vn = start_of(lengths[d], lengths[d + R]), 0
                    for byte in text:
                        vp, vn = tiny_column(vp, vn, peq.get(byte, [0] * R)[d])
                    j, k = d % H, d // H
                    packed[j] |= (half_distance(len(text), vp, vn, 0) << (8 * k)) | (half_distance(len(text), vp, vn, 1) << (16 + 8 * k))
                for j in range(H):
                    out[j][column] = packed[j]
            else:  # kind A: a cluster of R lanes, lane d the register d
                for d in range(R):
                    vp, vn = start_of(lengths[d], lengths[d + R]), 0
                    for byte in text:
                        vp, vn = tiny_column(vp, vn, peq.get(byte, [0] * R)[d])
                    for half in (0, 1):
                        distance = half_distance(len(text), vp, vn, half)
                        assert distance <= 255
                        shift = 8 * (d // H + 2 * half)
                        out[d % H][column] = (out[d % H][column] & ~(0xFF << shift)) | (distance << shift)
        for s in range(GROUP):  # the rows leave: slot s reads byte s / H of out[s % H]
            if lengths[s] is None:
                continue
            for column in range(len(candidates)):
                results[group_first + s][column] = (out[s % H][column] >> (8 * (s // H))) & 0xFF
    if long_queries:  # kinds B and C, by the longest of them
        longest = max(len(queries[q]) for q in long_queries)
        words = 1 if longest <= 32 else 2 if longest <= 64 else 4 if longest <= 128 else 8
        for q in long_queries:
            for column, text in enumerate(candidates):
                results[q][column] = long_pattern_distance(queries[q], text, words)
    return results


ALPHABET = b"etaoinshrdlu" + bytes([0xC3, 0xA9, 0xFF, 0x80])


def _token(rng, longest, long_share):
    if longest > 16 and rng.random() < long_share:
        return bytes(rng.choice(ALPHABET) for _ in range(rng.choice([17, 31, 32, 33, longest, rng.randint(17, longest)])))
    return bytes(rng.choice(ALPHABET) for _ in range(rng.choice([0, 1, 2, 3, 5, 8, 15, 16, rng.randint(0, 16)])))


@pytest.mark.parametrize("rows,columns,longest,long_share,dense", [
    (5, 9, 16, 0.0, False), (33, 40, 16, 0.0, False), (40, 30, 40, 0.1, False), (20, 24, 64, 0.15, False), (12, 20, 128, 0.2, True),
    (6, 10, 255, 0.5, True), (35, 12, 70, 1.0, True),
])
def test_a_workgroups_share_scores_like_the_recurrence(rows, columns, longest, long_share, dense):
    rng = random.Random(rows * 1009 + columns * 31 + longest)
    queries = [_token(rng, longest, long_share) for _ in range(rows)]
    candidates = [_token(rng, longest, long_share) for _ in range(columns)]
    got = replay_workgroup(queries, candidates, dense=dense)
    if got is None:  # refused: more than a quarter long - only ever without the testing knob
        assert not dense
        return
    for q, pattern in enumerate(queries):
        for c, text in enumerate(candidates):
            assert got[q][c] == levenshtein(pattern, text), (q, c, pattern, text)


def test_the_halves_of_a_register_do_not_leak_into_each_other():
    """Two sixteen-byte patterns in one register against sixteen-byte texts: the low half's sum overflows its sixteen bits on most
    columns, its top row's deltas are shifted out - neither may reach the high half."""
    rng = random.Random(5)
    for _ in range(50):
        low, high = (bytes(rng.choice(b"ab") for _ in range(16)) for _ in range(2))
        text = bytes(rng.choice(b"ab") for _ in range(16))
        masks = {}
        for position, (x, y) in enumerate(zip(low, high)):
            masks[x] = masks.get(x, 0) | (1 << position)
            masks[y] = masks.get(y, 0) | (0x10000 << position)
        vp, vn = start_of(16, 16), 0
        for byte in text:
            vp, vn = tiny_column(vp, vn, masks.get(byte, 0))
        assert half_distance(16, vp, vn, 0) == levenshtein(low, text) and half_distance(16, vp, vn, 1) == levenshtein(high, text)


def test_dense_shares_are_refused_and_what_is_too_long_is_not_taken():
    word, url = b"word", b"u" * 40
    assert replay_workgroup([word] * 32, [word] * 191 + [url] * 65) is None  # more than a quarter of the block
    assert replay_workgroup([word] * 32, [word] * 192 + [url] * 64) is not None
    assert replay_workgroup([word] * 23 + [url] * 9, [word] * 8) is None  # more than a quarter of the span (and more than four)
    assert replay_workgroup([word] * 24 + [url] * 8, [word] * 8) is not None
    assert replay_workgroup([url] * 4, [word] * 8) is not None  # up to four long queries are always taken
    assert replay_workgroup([word], [b"x" * 256]) is None and replay_workgroup([b"x" * 255], [word], dense=True) is not None


def test_agrees_with_the_oracle(oracle):
    rng = random.Random(77)
    queries = [_token(rng, 48, 0.08) for _ in range(70)]
    candidates = [_token(rng, 48, 0.08) for _ in range(90)]
    got = replay_workgroup(queries, candidates)
    expected = oracle.levenshtein(queries, candidates)
    assert got is not None and [[int(x) for x in row] for row in expected] == got
