"""`-m "not gpu"`: the arithmetic of the one-launch kernel's bodies (hip/myers_queue.hip, hip/myers_core.hpp) restated in Python.

A pattern of m symbols lives RIGHT-ALIGNED in L lanes x W words of 32 bits: the 32 W L - m low rows are phantoms whose vertical
deltas start (and stay) at zero and whose match masks are empty - they pass the `+1` of DP row zero upwards unchanged, so no lane
needs to know where the pattern begins.  Lane k takes a column after lane k - 1 did and receives the two horizontal delta bits that
left that lane's top row (`myers_strip_column`: hp_in / hn_in; the head lane takes hp = 1).  The distance is the text's length plus
the sum over all words of popcount(VP) - popcount(VN): D[m][n] = D[0][n] + the vertical deltas of column n.  This file drives that
arithmetic word by word against the oracle's Levenshtein, for every body shape the kernel has (one lane of 1 ... 20 words; teams of
2 ... 16 lanes x 4 / 8 / 12 / 16 words), so that a change of the layout can be checked WITHOUT a GPU."""
import random

import pytest

MASK = 0xFFFFFFFF


def strip_column(vp, vn, eq, hp_in, hn_in):
    """`myers_strip_column`: one column of one lane's words (lists of 32-bit ints, updated in place); returns (hp_out, hn_out)."""
    carry, hp_below, hn_below = 0, 0, 0
    for w in range(len(vp)):
        xv = eq[w] | vn[w]
        eq_in = eq[w] | hn_in if w == 0 else eq[w]
        total = (eq_in & vp[w]) + vp[w] + carry
        carry, total = total >> 32, total & MASK
        d0 = (total ^ vp[w]) | eq_in
        hp = (vn[w] | ~(d0 | vp[w])) & MASK
        hn = vp[w] & d0
        hp_shifted = ((hp << 1) | (hp_in if w == 0 else hp_below >> 31)) & MASK
        hn_shifted = ((hn << 1) | (hn_in if w == 0 else hn_below >> 31)) & MASK
        hp_below, hn_below = hp, hn
        vp[w] = (hn_shifted | ~(xv | hp_shifted)) & MASK
        vn[w] = hp_shifted & xv
    return hp_below >> 31, hn_below >> 31


def body_distance(query, text, words_per_lane, lanes):
    """One (query, candidate) pair the way a team of `lanes` lanes x `words_per_lane` words scores it."""
    bits = 32 * words_per_lane * lanes
    assert len(query) <= bits
    pad = bits - len(query)
    masks = {}  # symbol -> list of words over the whole right-aligned pattern
    for i, symbol in enumerate(query):
        position = pad + i
        masks.setdefault(symbol, [0] * (words_per_lane * lanes))[position >> 5] |= 1 << (position & 31)
    empty = [0] * (words_per_lane * lanes)
    vp = [[0] * words_per_lane for _ in range(lanes)]
    vn = [[0] * words_per_lane for _ in range(lanes)]
    for part in range(lanes):
        for w in range(words_per_lane):
            first_bit = 32 * (part * words_per_lane + w)
            vp[part][w] = MASK if first_bit >= pad else 0 if first_bit + 32 <= pad else (MASK << (pad - first_bit)) & MASK
    for symbol in text:  # the lanes' skew (lane k one column behind lane k - 1) does not change what a lane computes
        row = masks.get(symbol, empty)
        hp_in, hn_in = 1, 0  # the head lane: DP row zero grows by one per column
        for part in range(lanes):
            eq = row[part * words_per_lane:(part + 1) * words_per_lane]
            hp_in, hn_in = strip_column(vp[part], vn[part], eq, hp_in, hn_in)
    delta = sum(bin(word).count("1") for lane in vp for word in lane) - sum(bin(word).count("1") for lane in vn for word in lane)
    return len(text) + delta


def levenshtein(a, b):
    previous = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        current = [i]
        for j, y in enumerate(b, 1):
            current.append(min(previous[j] + 1, current[j - 1] + 1, previous[j - 1] + (x != y)))
        previous = current
    return previous[-1]


SHAPES = [(words, 1) for words in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20)] + \
         [(words, lanes) for words in (4, 8, 12, 16) for lanes in (2, 3, 5, 7, 9, 12, 16) if words * lanes <= 64]


@pytest.mark.parametrize("words,lanes", SHAPES)
def test_every_body_shape_scores_like_the_recurrence(words, lanes):
    rng = random.Random(words * 100 + lanes)
    capacity = 32 * words * lanes
    for length in sorted({0, 1, 31, 32, 33, capacity - 33, capacity - 32, capacity - 1, capacity, rng.randint(0, capacity), rng.randint(0, capacity)}):
        if length < 0 or length > capacity:
            continue
        query = bytes(rng.choice(b"ACGT") for _ in range(length))
        for text_length in (0, 1, 40, rng.randint(2, 120)):
            text = bytes(rng.choice(b"ACGT") for _ in range(text_length))
            if rng.random() < 0.5 and length and text_length:  # related strings: long runs of matches, the carry chain at work
                text = (query * (text_length // max(length, 1) + 1))[: text_length]
            assert body_distance(query, text, words, lanes) == levenshtein(query, text), (words, lanes, length, text_length)


def test_agrees_with_the_oracle_on_longer_pairs(oracle):
    rng = random.Random(5)
    queries = [bytes(rng.choice(b"ab") for _ in range(n)) for n in (700, 1025, 2048)]
    texts = [bytes(rng.choice(b"ab") for _ in range(n)) for n in (300, 900)]
    expected = oracle.levenshtein(queries, texts)
    for q, query in enumerate(queries):
        for t, text in enumerate(texts):
            assert body_distance(query, text, 4, 16) == int(expected[q, t])
            assert body_distance(query, text, 16, 4) == int(expected[q, t])
