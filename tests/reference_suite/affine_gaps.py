"""A stand-in for the `affine_gaps` PyPI package, which the reference's own test suite (`test/similarities.py:27`) imports as
its Needleman-Wunsch / Smith-Waterman baseline and which cannot be installed here (no network).

TEST INFRASTRUCTURE.  It offers exactly the names that test file uses and answers them with this repository's CPU oracle
(`oracle/sz_oracle.c`, pinned against the reference's own engines: DESIGN.md section 2) - so when
`tests/test_reference_suite.py` runs the reference's tests, "the `affine_gaps` baseline" is the oracle.  The substitution
matrix is BLOSUM62 as the reference itself tabulates it (`serial.hpp:221-287`, via `stringzilla_amd.matrices`); gap costs
are the reference bench's affine pair (`bench/similarities.cuh:648`).  Any consistent choice would do: the tests feed the
same alphabet, matrix and gaps to the engine under test and to this baseline.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import binding as _binding  # noqa: E402

_byte_to_class, _class_costs = _binding.blosum62()
_letters = [letter for letter in "ARNDCQEGHILKMFPSTWYVBZX" if _byte_to_class[ord(letter)] or letter == "A"]
default_proteins_alphabet = "".join(_letters)
default_proteins_matrix = np.array(
    [[int(_class_costs[_byte_to_class[ord(a)], _byte_to_class[ord(b)]]) for b in _letters] for a in _letters], dtype=np.int8)
default_gap_opening = -4
default_gap_extension = -1


def _tables(substitution_alphabet, substitution_matrix):
    byte_to_class = np.zeros(256, dtype=np.uint8)
    class_costs = np.zeros((32, 32), dtype=np.int8)
    count = len(substitution_alphabet)
    byte_to_class[np.frombuffer(substitution_alphabet.encode(), dtype=np.uint8)] = np.arange(1, count + 1, dtype=np.uint8)
    class_costs[1:count + 1, 1:count + 1] = np.asarray(substitution_matrix)[:count, :count]
    return byte_to_class, class_costs


def _encode(text):
    return text.encode() if isinstance(text, str) else bytes(text)


def needleman_wunsch_gotoh_score(first, second, substitution_alphabet=default_proteins_alphabet,
                                 substitution_matrix=default_proteins_matrix, gap_opening=default_gap_opening,
                                 gap_extension=default_gap_extension, **_):
    scores = _binding.oracle().needleman_wunsch([_encode(first)], [_encode(second)], *_tables(substitution_alphabet, substitution_matrix),
                                                gap_opening, gap_extension)
    return int(scores[0, 0])


def smith_waterman_gotoh_score(first, second, substitution_alphabet=default_proteins_alphabet,
                               substitution_matrix=default_proteins_matrix, gap_opening=default_gap_opening,
                               gap_extension=default_gap_extension, **_):
    scores = _binding.oracle().smith_waterman([_encode(first)], [_encode(second)], *_tables(substitution_alphabet, substitution_matrix),
                                              gap_opening, gap_extension)
    return int(scores[0, 0])


def needleman_wunsch_gotoh(first, second, **_):
    """Only used by the reference's tests to decorate a failure message; the stand-in has no traceback to offer."""
    return str(first), str(second)


smith_waterman_gotoh = needleman_wunsch_gotoh
