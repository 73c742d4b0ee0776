"""ctypes access to the CHECKERS: the plain-C oracle (oracle/libsz_oracle.so) and, when built, the real reference
engines (oracle/_ref/libszs_ref.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package `stringzilla_amd` never imports this module.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libsz_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libszs_ref.so")

_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_i8 = ctypes.c_int8


def build(with_reference: bool = True) -> None:
    """Compile the checkers (gcc for the restatement; g++ over /root/reference for _ref when that tree exists)."""
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    if with_reference and os.path.isdir("/root/reference/include/stringzillas"):
        stale = not os.path.exists(_REF_SO) or os.path.getmtime(_REF_SO) < os.path.getmtime(
            os.path.join(_HERE, "ref_shim.cpp")
        )
        if stale:
            subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def make_tape(strings: Sequence[bytes], offset_dtype=np.uint64):
    """Packs byte strings into an Arrow-like tape: (contiguous uint8 array, count+1 offsets)."""
    lengths = np.fromiter((len(s) for s in strings), dtype=np.int64, count=len(strings))
    offsets = np.zeros(len(strings) + 1, dtype=offset_dtype)
    np.cumsum(lengths, out=offsets[1:])
    data = np.frombuffer(b"".join(strings), dtype=np.uint8).copy() if len(strings) else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)  # keep a valid pointer for empty corpora
    return data, offsets


def _ptr(array: Optional[np.ndarray]):
    return None if array is None else array.ctypes.data_as(_vp)


class _Checker:
    """Shared calling convention of the oracle and the reference shim: u64 tapes in, dense matrix out."""

    def __init__(self, lib, prefix: str, extra_args: tuple):
        self._lib = lib
        self._prefix = prefix
        self._extra = extra_args

    def _run(self, name, cost_args, queries, candidates, dtype):
        q_data, q_off = make_tape(queries)
        symmetric = candidates is None
        if symmetric:
            c_data, c_off, c_count = None, None, len(queries)
        else:
            c_data, c_off = make_tape(candidates)
            c_count = len(candidates)
        results = np.zeros((len(queries), c_count), dtype=dtype)
        fn = getattr(self._lib, self._prefix + name)
        fn.restype = ctypes.c_int if self._prefix.startswith("szs_ref") else None
        tape_args = [_ptr(q_data), _ptr(q_off), _sz(len(queries)), _ptr(c_data), _ptr(c_off), _sz(c_count)]
        tail = [_ptr(results), _sz(max(c_count, 1))]
        if self._prefix.startswith("szs_ref"):  # shim: (tier, threads, costs..., tapes..., results, stride)
            status = fn(*self._extra, *cost_args, *tape_args, *tail)
            if status != 0:
                raise RuntimeError(f"reference engine returned status {status}")
        else:  # oracle: (tapes..., costs..., results, stride)
            fn(*tape_args, *cost_args, *tail)
        return results

    def levenshtein(self, queries, candidates=None, match=0, mismatch=1, open=1, extend=1):
        name = "levenshtein" if self._prefix.startswith("szs_ref") else "levenshtein_cross"
        return self._run(name, [_i8(match), _i8(mismatch), _i8(open), _i8(extend)], queries, candidates, np.uint64)

    def levenshtein_utf8(self, queries, candidates=None, match=0, mismatch=1, open=1, extend=1):
        name = "levenshtein_utf8" if self._prefix.startswith("szs_ref") else "levenshtein_utf8_cross"
        return self._run(name, [_i8(match), _i8(mismatch), _i8(open), _i8(extend)], queries, candidates, np.uint64)

    def _scores(self, kind, queries, candidates, byte_to_class, class_costs, open, extend):
        byte_to_class = np.ascontiguousarray(byte_to_class, dtype=np.uint8)
        class_costs = np.ascontiguousarray(class_costs, dtype=np.int8).reshape(-1)
        assert byte_to_class.size == 256 and class_costs.size == 1024
        name = kind if self._prefix.startswith("szs_ref") else kind + "_cross"
        return self._run(
            name, [_ptr(byte_to_class), _ptr(class_costs), _i8(open), _i8(extend)], queries, candidates, np.int64
        )

    def needleman_wunsch(self, queries, candidates, byte_to_class, class_costs, open=-1, extend=-1):
        return self._scores("needleman_wunsch", queries, candidates, byte_to_class, class_costs, open, extend)

    def smith_waterman(self, queries, candidates, byte_to_class, class_costs, open=-1, extend=-1):
        return self._scores("smith_waterman", queries, candidates, byte_to_class, class_costs, open, extend)


_oracle_lib = None
_ref_lib = None


def oracle() -> _Checker:
    """The plain-C restatement.  Built on demand (gcc is available everywhere this runs)."""
    global _oracle_lib
    if _oracle_lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(with_reference=False)
        _oracle_lib = ctypes.CDLL(_ORACLE_SO)
        for name in ("szo_levenshtein", "szo_levenshtein_linear", "szo_levenshtein_affine", "szo_levenshtein_myers",
                     "szo_levenshtein_utf8"):
            getattr(_oracle_lib, name).restype = ctypes.c_uint64
        for name in ("szo_needleman_wunsch", "szo_smith_waterman"):
            getattr(_oracle_lib, name).restype = ctypes.c_int64
        _oracle_lib.szo_worst_case_reach.restype = ctypes.c_uint64
    return _Checker(_oracle_lib, "szo_", ())


def oracle_lib():
    oracle()
    return _oracle_lib


def reference_available() -> bool:
    return os.path.exists(_REF_SO)


def reference(tier: int = 0, threads: int = 1) -> _Checker:
    """The real reference engines (tier 0 serial, 1 Haswell, 2 Ice Lake; clamped to what the host CPU supports)."""
    global _ref_lib
    if _ref_lib is None:
        if not os.path.exists(_REF_SO):
            raise FileNotFoundError(f"{_REF_SO} is not built (needs /root/reference; run `make -C oracle ref`)")
        _ref_lib = ctypes.CDLL(_REF_SO)
    return _Checker(_ref_lib, "szs_ref_", (ctypes.c_int(tier), ctypes.c_int(threads)))


def reference_best_tier() -> int:
    reference()
    return int(_ref_lib.szs_ref_best_tier())


def blosum62():
    lib = oracle_lib()
    byte_to_class, class_costs = np.zeros(256, np.uint8), np.zeros(1024, np.int8)
    lib.szo_blosum62(_ptr(byte_to_class), _ptr(class_costs))
    return byte_to_class, class_costs.reshape(32, 32)


def nuc44():
    lib = oracle_lib()
    byte_to_class, class_costs = np.zeros(256, np.uint8), np.zeros(1024, np.int8)
    lib.szo_nuc44(_ptr(byte_to_class), _ptr(class_costs))
    return byte_to_class, class_costs.reshape(32, 32)


def reference_table(which: int):
    reference()
    byte_to_class, class_costs = np.zeros(256, np.uint8), np.zeros(1024, np.int8)
    _ref_lib.szs_ref_substitution_table(ctypes.c_int(which), _ptr(byte_to_class), _ptr(class_costs))
    return byte_to_class, class_costs.reshape(32, 32)


# ---- rolling MinHash / Count-Min fingerprints ----------------------------------------------------------------------------


def _fingerprint_arguments(texts, dimensions, window_widths, seed):
    data, offsets = make_tape(list(texts))
    widths = None if window_widths is None else np.ascontiguousarray(window_widths, dtype=np.uint64)
    hashes = np.zeros((len(texts), dimensions), dtype=np.uint32)
    counts = np.zeros((len(texts), dimensions), dtype=np.uint32)
    return data, offsets, widths, hashes, counts


def oracle_fingerprints(texts: Sequence[bytes], dimensions: int, window_widths=None, seed: int = 0, alphabet_size: int = 256):
    """(min_hashes, min_counts) of the plain-C restatement (oracle/sz_oracle_fingerprints.c)."""
    lib = oracle_lib()
    data, offsets, widths, hashes, counts = _fingerprint_arguments(texts, dimensions, window_widths, seed)
    lib.szo_fingerprints_cross(_ptr(data), _ptr(offsets), _sz(len(texts)), _sz(dimensions), _sz(alphabet_size), _ptr(widths),
                               _sz(0 if widths is None else len(widths)), ctypes.c_uint64(seed), _ptr(hashes), _ptr(counts))
    return hashes, counts


def reference_fingerprints(texts: Sequence[bytes], dimensions: int, window_widths=None, seed: int = 0, alphabet_size: int = 256):
    """The same through the REFERENCE's own serial engines (oracle/_ref/libszs_ref.so), composed like its C shim does;
    also returns which engine that was: 1 = 64-dimension slices, 2 = per-dimension fallback."""
    lib = ctypes.CDLL(_REF_SO)
    data, offsets, widths, hashes, counts = _fingerprint_arguments(texts, dimensions, window_widths, seed)
    kind = lib.szs_ref_fingerprints(_sz(dimensions), _sz(alphabet_size), _ptr(widths), _sz(0 if widths is None else len(widths)),
                                    ctypes.c_uint64(seed), _ptr(data), _ptr(offsets), _sz(len(texts)), _ptr(hashes), _ptr(counts))
    if kind <= 0:
        raise RuntimeError("reference fingerprint engine failed")
    return hashes, counts, kind


def reference_fingerprints_tiered(texts: Sequence[bytes], dimensions: int, tier: int, threads: int, window_widths=None, seed: int = 0,
                                  alphabet_size: int = 256):
    """(min_hashes, min_counts, tier that ran) through the reference's SIMD hashers (Skylake / Haswell / serial slices of 64
    dimensions), the texts dealt over `threads` host threads: the CPU baseline of bench.py's fingerprints record."""
    lib = ctypes.CDLL(_REF_SO)
    data, offsets, widths, hashes, counts = _fingerprint_arguments(texts, dimensions, window_widths, seed)
    ran = lib.szs_ref_fingerprints_tiered(ctypes.c_int(tier), ctypes.c_int(threads), _sz(dimensions), _sz(alphabet_size), _ptr(widths),
                                          _sz(0 if widths is None else len(widths)), ctypes.c_uint64(seed), _ptr(data), _ptr(offsets),
                                          _sz(len(texts)), _ptr(hashes), _ptr(counts))
    if ran < 0:
        raise RuntimeError("reference fingerprint engine failed")
    return hashes, counts, int(ran)
