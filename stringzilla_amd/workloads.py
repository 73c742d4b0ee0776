"""Synthetic inputs for the configurations named in BASELINE.json / SURVEY.md section 8(d).

Pure numpy, seeded, no oracle and no device code in here: used by bench.py, by the GPU parity tests and by the CPU
baseline leg so that all three see byte-identical batches.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import Strs

ASCII_PRINTABLE = np.arange(0x20, 0x7F, dtype=np.uint8)
AMINO_ACIDS = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
NUCLEOTIDES = np.frombuffer(b"ACGT", dtype=np.uint8)
MT19937_64_LIBRARY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libszs_workloads_mt19937.so")


def random_tape(rng: np.random.Generator, count: int, low: int, high: int, alphabet: np.ndarray) -> Strs:
    """`count` strings with lengths ~ U[low, high] over `alphabet`, packed as a u32 tape."""
    lengths = rng.integers(low, high + 1, size=count, dtype=np.int64)
    offsets = np.zeros(count + 1, dtype=np.uint32)
    np.cumsum(lengths, out=offsets[1:])
    data = alphabet[rng.integers(0, len(alphabet), size=int(offsets[-1]))]
    return Strs.from_tape(data, offsets)


def zipf_utf8_tape(rng: np.random.Generator, count: int, low: int = 8, high: int = 2048, exponent: float = 1.1) -> Strs:
    """Config 5: mixed 1-4-byte UTF-8 text whose BYTE length follows Zipf(s) clipped to [low, high]."""
    ranks = np.arange(low, high + 1, dtype=np.float64)
    weights = ranks ** (-exponent)
    lengths = rng.choice(np.arange(low, high + 1), size=count, p=weights / weights.sum())
    pools = {
        1: [bytes([c]) for c in range(0x20, 0x7F)],
        2: [chr(c).encode() for c in list(range(0xC0, 0x17F)) + list(range(0x410, 0x450))],
        3: [chr(c).encode() for c in range(0x4E00, 0x4E00 + 512)],
        4: [chr(c).encode() for c in range(0x1F600, 0x1F640)],
    }
    widths = rng.choice([1, 2, 3, 4], size=int(lengths.sum()), p=[0.70, 0.20, 0.08, 0.02])
    strings, cursor = [], 0
    for target in lengths:
        parts, size = [], 0
        while size < target:
            width = int(widths[cursor % len(widths)])
            cursor += 1
            if size + width > target:
                width = 1  # finish with ASCII so the byte length is exact
            pool = pools[width]
            parts.append(pool[int(rng.integers(0, len(pool)))])
            size += width
        strings.append(b"".join(parts))
    return Strs(strings)


def mt19937_64_tape(seed: int, count: int, low: int, high: int, alphabet: np.ndarray) -> Strs:
    """The same kind of tape from `std::mt19937_64` (csrc/workloads/workloads_mt19937.cpp spells the mapping out), for callers
    that want to reproduce a batch from C++: SURVEY.md section 8(d) names that generator.  Raises when the helper library
    (built by csrc/Makefile beside the scoring library) is missing: a batch of another generator under the same name is no
    substitute."""
    import ctypes

    if not os.path.exists(MT19937_64_LIBRARY):
        raise FileNotFoundError(f"{MT19937_64_LIBRARY} is not built (make -C stringzilla_amd/csrc)")
    fill = ctypes.CDLL(MT19937_64_LIBRARY).szs_workload_mt19937_64
    fill.restype = ctypes.c_uint64
    fill.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    alphabet = np.ascontiguousarray(alphabet, dtype=np.uint8)
    offsets = np.zeros(count + 1, dtype=np.uint32)
    total = fill(seed, count, low, high, alphabet.ctypes.data, len(alphabet), offsets.ctypes.data, None)
    data = np.empty(int(total), dtype=np.uint8)
    fill(seed, count, low, high, alphabet.ctypes.data, len(alphabet), offsets.ctypes.data, data.ctypes.data)
    return Strs.from_tape(data, offsets)


@dataclass
class Workload:
    name: str
    kind: str  # "levenshtein" | "levenshtein_utf8" | "needleman_wunsch" | "smith_waterman"
    queries: Strs
    candidates: Strs
    costs: dict  # engine constructor keywords besides the substitution table
    table: Optional[str] = None  # "blosum62" | "nuc44"

    @property
    def pairs(self) -> int:
        return len(self.queries) * len(self.candidates)

    @property
    def cells(self) -> int:
        return int(self.queries.lengths().sum()) * int(self.candidates.lengths().sum())


def config(index: int, scale: float = 1.0, generator: str = "numpy") -> Workload:
    """The five BASELINE.json configs as concrete, seeded batches.  `scale` < 1 shrinks
    the matrix side for parity tests that must finish in seconds on the CPU oracle.  `generator`: "numpy" (the committed
    profiles and checksums) or "mt19937_64" (configs 1-4: the same shapes from `std::mt19937_64`, reproducible from C++)."""
    rng = np.random.default_rng(index)
    side = lambda n: max(1, int(round(n * scale)))
    if generator == "mt19937_64" and index in (1, 2, 3, 4, 9):
        shape = {1: (100, 48, 80, ASCII_PRINTABLE), 2: (1024, 96, 160, ASCII_PRINTABLE), 3: (1024, 384, 640, AMINO_ACIDS),
                 4: (512, 3072, 5120, NUCLEOTIDES), 9: (1024, 128, 128, ASCII_PRINTABLE)}[index]
        template = config(index, 0.01)  # names, kinds and costs of the numpy variant
        return Workload(template.name + " [std::mt19937_64]", template.kind,
                        mt19937_64_tape(1000 * index + 0, side(shape[0]), shape[1], shape[2], shape[3]),
                        mt19937_64_tape(1000 * index + 1, side(shape[0]), shape[1], shape[2], shape[3]), template.costs, template.table)
    if generator != "numpy":
        raise ValueError(f"generator {generator!r} has no config {index}")
    if index == 1:
        return Workload("cfg1: 100x100 ASCII len U[48,80], Levenshtein unit", "levenshtein",
                        random_tape(rng, side(100), 48, 80, ASCII_PRINTABLE),
                        random_tape(rng, side(100), 48, 80, ASCII_PRINTABLE), dict(match=0, mismatch=1, open=1, extend=1))
    if index == 2:
        return Workload("cfg2: 1024x1024 ASCII len U[96,160], Levenshtein unit", "levenshtein",
                        random_tape(rng, side(1024), 96, 160, ASCII_PRINTABLE),
                        random_tape(rng, side(1024), 96, 160, ASCII_PRINTABLE), dict(match=0, mismatch=1, open=1, extend=1))
    if index == 3:
        return Workload("cfg3: 1024x1024 protein len U[384,640], NW BLOSUM62 linear -4", "needleman_wunsch",
                        random_tape(rng, side(1024), 384, 640, AMINO_ACIDS),
                        random_tape(rng, side(1024), 384, 640, AMINO_ACIDS), dict(open=-4, extend=-4), "blosum62")
    if index == 4:
        return Workload("cfg4: 512x512 DNA len U[3072,5120], SW NUC.4.4 affine -4/-1", "smith_waterman",
                        random_tape(rng, side(512), 3072, 5120, NUCLEOTIDES),
                        random_tape(rng, side(512), 3072, 5120, NUCLEOTIDES), dict(open=-4, extend=-1), "nuc44")
    if index == 5:
        return Workload("cfg5: 3163x3163 UTF-8 Zipf(1.1) bytes [8,2048], byte-level Levenshtein unit", "levenshtein",
                        zipf_utf8_tape(rng, side(3163)), zipf_utf8_tape(rng, side(3163)),
                        dict(match=0, mismatch=1, open=1, extend=1))
    if index == 6:  # config 5's batch scored at the CODEPOINT level: not a BASELINE.json line
        rng = np.random.default_rng(5)
        return Workload("cfg5u: 3163x3163 UTF-8 Zipf(1.1) bytes [8,2048], codepoint-level Levenshtein unit",
                        "levenshtein_utf8", zipf_utf8_tape(rng, side(3163)), zipf_utf8_tape(rng, side(3163)),
                        dict(match=0, mismatch=1, open=1, extend=1))
    if index == 9:  # config 2's "peak" variant: every string exactly 128 bytes - no ragged lanes, no phantom rows
        rng = np.random.default_rng(2)
        return Workload("cfg2p: 1024x1024 ASCII len 128 exactly, Levenshtein unit", "levenshtein",
                        random_tape(rng, side(1024), 128, 128, ASCII_PRINTABLE), random_tape(rng, side(1024), 128, 128, ASCII_PRINTABLE),
                        dict(match=0, mismatch=1, open=1, extend=1))
    if index == 10:  # tiny tokens: what the reference's own benchmark tokeniser feeds (words of text, bench/shared.hpp:240-290) - the
        # regime of its direct kernel (cuda.cuh:2864).  4096 x 4096 word-like tokens: lengths 1 ... 16 with the shape of English
        # word lengths (mean ~5.6) and one token in 25 between 17 and 48 bytes (identifiers, paths), letters by their frequency
        rng = np.random.default_rng(10)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        weights = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.2, 0.2, 0.1, 0.1])
        def words(count):
            lengths = np.clip(rng.poisson(4.6, size=count) + 1, 1, 16)
            long_ones = rng.random(count) < 0.04
            lengths[long_ones] = rng.integers(17, 49, size=int(long_ones.sum()))
            offsets = np.zeros(count + 1, dtype=np.uint32)
            np.cumsum(lengths, out=offsets[1:])
            return Strs.from_tape(letters[rng.choice(len(letters), size=int(offsets[-1]), p=weights / weights.sum())], offsets)
        return Workload("words: 4096x4096 word-like tokens (mean 6.6 bytes, 4 % of 17-48), Levenshtein unit", "levenshtein",
                        words(side(4096)), words(side(4096)), dict(match=0, mismatch=1, open=1, extend=1))
    if index == 12:  # config 10's words at the CODEPOINT level: the same lengths in runes, one letter in twelve an accented or Cyrillic one
        # (two bytes of UTF-8) - through `szs_levenshtein_distances_utf8_*`, the regime of the reference's per-thread kernel for short
        # runes (cuda.cuh:3294)
        rng = np.random.default_rng(12)
        plain, accented = "etaoinshrdlcumwfgypbvkjxqz", "éèüöäßñçåøæîôдежзиклмноп"
        def words(count):
            lengths = np.clip(rng.poisson(4.6, size=count) + 1, 1, 16)
            long_ones = rng.random(count) < 0.04
            lengths[long_ones] = rng.integers(17, 49, size=int(long_ones.sum()))
            total = int(lengths.sum())
            runes = np.where(rng.random(total) < 1 / 12, np.array(list(accented))[rng.integers(0, len(accented), size=total)],
                             np.array(list(plain))[rng.integers(0, len(plain), size=total)])
            ends = np.cumsum(lengths)
            return Strs(["".join(runes[end - length:end]).encode() for end, length in zip(ends, lengths)])
        return Workload("wordsu: 4096x4096 word-like tokens (mean 6.6 runes, 1 in 12 of two bytes), codepoint-level Levenshtein unit",
                        "levenshtein_utf8", words(side(4096)), words(side(4096)), dict(match=0, mismatch=1, open=1, extend=1))
    if index in (7, 8):  # config 2's batch under NON-UNIT costs (a north_star function: szs_levenshtein_distances_init takes all four):
        rng = np.random.default_rng(2)  # 7: linear gaps, match 1 / mismatch 3 / gap 3; 8: affine gaps, mismatch 1 / open 4 / extend 2
        costs = dict(match=1, mismatch=3, open=3, extend=3) if index == 7 else dict(match=0, mismatch=1, open=4, extend=2)
        gaps = "linear 1/3/3" if index == 7 else "affine 0/1/4/2"
        return Workload(f"cfg2w{'l' if index == 7 else 'a'}: 1024x1024 ASCII len U[96,160], Levenshtein {gaps}", "levenshtein",
                        random_tape(rng, side(1024), 96, 160, ASCII_PRINTABLE), random_tape(rng, side(1024), 96, 160, ASCII_PRINTABLE), costs)
    raise ValueError(f"unknown config {index}")


def tokenize_dataset(data: bytes, tokens: str = "words", max_tokens: int = 0, unique: bool = False) -> Strs:
    """The reference's benchmark tokeniser (`bench/shared.hpp:240-262,440-480`), for runs on real text (`xlsum.csv`,
    `acgt_*.txt`: `CONTRIBUTING.md:320-324`) instead of the synthetic configurations:

      - the dataset is cut to the largest power of two that fits (`bit_floor`, `:451`);
      - `tokens`: "file" (one token), "lines" (split at `\\n`), "words" (split at C `isspace` bytes: space, \\t \\n \\v \\f \\r)
        or a positive integer N as a string (words of exactly N bytes) - `STRINGWARS_TOKENS`; empty tokens are dropped;
      - `unique` sorts and deduplicates (`STRINGWARS_UNIQUE`), `max_tokens` keeps the first so many (`STRINGWARS_MAX_TOKENS`).

    Returns the tokens as a u32 tape over ONE copy of the dataset bytes (tokens are views, as in the reference).
    """
    size = 1 << (len(data).bit_length() - 1) if data else 0
    buffer = np.frombuffer(data, dtype=np.uint8, count=size)
    if tokens == "file":
        spans = [(0, size)] if size else []
    else:
        if tokens == "lines":
            separators = buffer == 0x0A
        else:
            separators = np.isin(buffer, np.frombuffer(b" \t\n\v\f\r", dtype=np.uint8))
        edges = np.flatnonzero(separators)
        starts = np.concatenate(([0], edges + 1))
        ends = np.concatenate((edges, [size]))
        keep = ends > starts
        if tokens not in ("lines", "words"):
            width = int(tokens)
            if width <= 0:
                raise ValueError("The tokenization mode must be 'file', 'lines', 'words', or a positive integer.")
            keep &= (ends - starts) == width
        spans = list(zip(starts[keep].tolist(), ends[keep].tolist()))
    pieces = [bytes(buffer[start:end]) for start, end in spans]
    if unique:
        pieces = sorted(set(pieces))
    if max_tokens:
        pieces = pieces[:max_tokens]
    return Strs(pieces)
