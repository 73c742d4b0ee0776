"""ctypes declarations of the C-ABI exported by `libstringzillas_rocm_shared.so` (include/stringzillas/*.h).

This is the reference-side binding a maintainer would write for the ROCm slot: the reference's CPython module
(/root/reference/python/stringzillas/similarities.c:235-428) calls exactly these functions with exactly these
argument lists.  There is no fallback: if the shared library (and the gfx950 code object inside it) is missing,
importing this module raises.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.environ.get("STRINGZILLAS_ROCM_LIBRARY", os.path.join(_HERE, "lib", "libstringzillas_rocm_shared.so"))

c_size_t, c_void_p, c_char_p, c_int = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int
c_int8 = ctypes.c_int8

# sz_status_t (include/stringzillas/stringzillas.h; reference include/stringzilla/types.h:810-833)
STATUS_NAMES = {
    0: "success", -10: "bad_alloc", -12: "invalid_utf8", -13: "contains_duplicates", -14: "overflow_risk",
    -15: "unexpected_dimensions", -16: "missing_gpu", -17: "device_code_mismatch", -18: "device_memory_mismatch",
    -19: "authentication_failed", -1: "unknown",
}

# sz_capability_t bits this build reports or reads
CAP_SERIAL = 1
CAP_PARALLEL = 1 << 2
CAP_CUDA = 1 << 21  # "a GPU engine exists" - the bit the ROCm build reports
CAPS_CPUS = 0x011FFCFD
CAPS_CUDA = (1 << 21) | (1 << 22) | (1 << 23)


class U32Tape(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("offsets", c_void_p), ("count", c_size_t)]


class U64Tape(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("offsets", c_void_p), ("count", c_size_t)]


MEMBER_START = ctypes.CFUNCTYPE(c_void_p, c_void_p, c_size_t)
MEMBER_LENGTH = ctypes.CFUNCTYPE(c_size_t, c_void_p, c_size_t)


class Sequence(ctypes.Structure):
    _fields_ = [("handle", c_void_p), ("count", c_size_t), ("get_start", MEMBER_START), ("get_length", MEMBER_LENGTH)]


class CallProfile(ctypes.Structure):
    _fields_ = [
        ("kernel_milliseconds", ctypes.c_double), ("host_milliseconds", ctypes.c_double),
        ("cells", ctypes.c_uint64), ("pairs", ctypes.c_uint64), ("algorithmic_bytes", ctypes.c_uint64),
        ("unique_bytes", ctypes.c_uint64), ("launches", ctypes.c_uint32), ("longest_query", ctypes.c_uint32),
        ("longest_candidate", ctypes.c_uint32), ("tier", ctypes.c_uint32), ("transposed", ctypes.c_uint32), ("cell_bits", ctypes.c_uint32),
        ("planner", ctypes.c_uint32), ("team", ctypes.c_uint32), ("team_wide", ctypes.c_uint32), ("streams", ctypes.c_uint32),
        ("queue_items", ctypes.c_uint32), ("queue_tiles", ctypes.c_uint32),
    ]


NODE_MOST_GPUS = 16


class NodeStats(ctypes.Structure):
    _fields_ = [
        ("gpus", c_size_t), ("wall_milliseconds", ctypes.c_double),
        ("busy_milliseconds", ctypes.c_double * NODE_MOST_GPUS), ("kernel_milliseconds", ctypes.c_double * NODE_MOST_GPUS),
        ("cells", ctypes.c_uint64 * NODE_MOST_GPUS), ("row_weights", ctypes.c_uint64 * NODE_MOST_GPUS),
        ("rows", ctypes.c_uint32 * NODE_MOST_GPUS),
        ("peer_copies", ctypes.c_uint32 * NODE_MOST_GPUS), ("staged_copies", ctypes.c_uint32 * NODE_MOST_GPUS),
        ("peer_pairs", ctypes.c_uint32), ("symmetric", ctypes.c_uint32),
    ]


ERR = ctypes.POINTER(c_char_p)
ENGINE_OUT = ctypes.POINTER(c_void_p)

# name -> (restype, argtypes); every symbol of include/stringzillas/stringzillas.h and stringzillas_rocm.h
_LEV_INIT = (c_int, [c_int8, c_int8, c_int8, c_int8, c_void_p, c_int, ENGINE_OUT, ERR])
_SCORE_INIT = (c_int, [c_void_p, c_void_p, c_int8, c_int8, c_void_p, c_int, ENGINE_OUT, ERR])
_CALL = (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, ERR])
_FP_CALL = (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, ERR])

SIGNATURES = {
    "szs_version_major": (c_int, []), "szs_version_minor": (c_int, []), "szs_version_patch": (c_int, []),
    "szs_capabilities_comptime": (c_int, []), "szs_capabilities_runtime": (c_int, []), "szs_capabilities": (c_int, []),
    "sz_memory_allocator_init_unified": (c_int, [c_void_p, ERR]),
    "szs_unified_alloc": (c_void_p, [c_size_t]), "szs_unified_free": (None, [c_void_p, c_size_t]),
    "szs_device_scope_init_default": (c_int, [ENGINE_OUT, ERR]),
    "szs_device_scope_init_cpu_cores": (c_int, [c_size_t, ENGINE_OUT, ERR]),
    "szs_device_scope_init_gpu_device": (c_int, [c_size_t, ENGINE_OUT, ERR]),
    "szs_device_scope_get_cpu_cores": (c_int, [c_void_p, ctypes.POINTER(c_size_t), ERR]),
    "szs_device_scope_get_gpu_device": (c_int, [c_void_p, ctypes.POINTER(c_size_t), ERR]),
    "szs_device_scope_get_capabilities": (c_int, [c_void_p, ctypes.POINTER(c_int), ERR]),
    "szs_device_scope_free": (None, [c_void_p]),
    "szs_levenshtein_distances_init": _LEV_INIT, "szs_levenshtein_distances": _CALL,
    "szs_levenshtein_distances_u32tape": _CALL, "szs_levenshtein_distances_u64tape": _CALL,
    "szs_levenshtein_distances_free": (None, [c_void_p]),
    "szs_levenshtein_distances_utf8_init": _LEV_INIT, "szs_levenshtein_distances_utf8": _CALL,
    "szs_levenshtein_distances_utf8_u32tape": _CALL, "szs_levenshtein_distances_utf8_u64tape": _CALL,
    "szs_levenshtein_distances_utf8_free": (None, [c_void_p]),
    "szs_needleman_wunsch_scores_init": _SCORE_INIT, "szs_needleman_wunsch_scores": _CALL,
    "szs_needleman_wunsch_scores_u32tape": _CALL, "szs_needleman_wunsch_scores_u64tape": _CALL,
    "szs_needleman_wunsch_scores_free": (None, [c_void_p]),
    "szs_smith_waterman_scores_init": _SCORE_INIT, "szs_smith_waterman_scores": _CALL,
    "szs_smith_waterman_scores_u32tape": _CALL, "szs_smith_waterman_scores_u64tape": _CALL,
    "szs_smith_waterman_scores_free": (None, [c_void_p]),
    "szs_fingerprints_init": (c_int, [c_size_t, c_size_t, c_void_p, c_size_t, ctypes.c_uint64, c_void_p, c_int, ENGINE_OUT, ERR]),
    "szs_fingerprints_sequence": _FP_CALL, "szs_fingerprints_u64tape": _FP_CALL, "szs_fingerprints_u32tape": _FP_CALL,
    "szs_fingerprints_free": (None, [c_void_p]),
    # ROCm-only additions (include/stringzillas/stringzillas_rocm.h)
    "szs_rocm_last_call_profile": (c_int, [c_void_p, ctypes.POINTER(CallProfile)]),
    "szs_rocm_shard_rows": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "szs_rocm_shard_triangle": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "szs_rocm_plan_probe": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "szs_rocm_orientation_probe": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "szs_rocm_team_orientation_probe": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "szs_rocm_launch_order_probe": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "szs_rocm_queue_probe": (c_int, [c_int, ctypes.c_uint32, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "szs_rocm_tuning_set": (c_int, [c_char_p, c_char_p]),
    "szs_rocm_team_shape": (ctypes.c_uint32, [c_size_t]),
    "szs_rocm_node_init": (c_int, [c_void_p, c_size_t, ENGINE_OUT, ERR]),
    "szs_rocm_node_size": (c_size_t, [c_void_p]), "szs_rocm_node_free": (None, [c_void_p]),
    "szs_rocm_node_levenshtein_distances_init": (c_int, [c_void_p, c_int8, c_int8, c_int8, c_int8, ENGINE_OUT, ERR]),
    "szs_rocm_node_levenshtein_distances_utf8_init": (c_int, [c_void_p, c_int8, c_int8, c_int8, c_int8, ENGINE_OUT, ERR]),
    "szs_rocm_node_needleman_wunsch_scores_init": (c_int, [c_void_p, c_void_p, c_void_p, c_int8, c_int8, ENGINE_OUT, ERR]),
    "szs_rocm_node_smith_waterman_scores_init": (c_int, [c_void_p, c_void_p, c_void_p, c_int8, c_int8, ENGINE_OUT, ERR]),
    "szs_rocm_node_engine_free": (None, [c_void_p]),
    "szs_rocm_node_scores_u32tape": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, ctypes.POINTER(NodeStats), ERR]),
    "szs_rocm_node_scores_u64tape": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, ctypes.POINTER(NodeStats), ERR]),
}

REFERENCE_SYMBOLS = [name for name in SIGNATURES if not name.startswith("szs_rocm_")]  # the reference's 41

if not os.path.exists(LIBRARY_PATH):
    raise ImportError(
        f"{LIBRARY_PATH} is missing: build it with `make -C stringzilla_amd/csrc` (or `python -c 'import "
        "__graft_entry__ as g; g.build()'`). stringzilla_amd has no CPU fallback."
    )

# ONE HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the SONAME this
# library NEEDs).  If torch is imported first, the dynamic linker binds this library to the runtime torch already loaded,
# so pointers, streams and device state are shared; if this library were loaded first, /opt/rocm's runtime would come in
# and torch would later load a SECOND one beside it, and then sees no GPU.  Hence: torch first, whenever it is installed.
try:
    import torch as _torch  # noqa: F401
except ImportError:  # C-only deployments: the RUNPATH'd /opt/rocm runtime is used
    _torch = None

lib = ctypes.CDLL(LIBRARY_PATH)
for _name, (_restype, _argtypes) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export what the header declares
    _fn.restype = _restype
    _fn.argtypes = _argtypes


class StringZillasError(RuntimeError):
    def __init__(self, status: int, message: str | None):
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"sz_status_t {self.status_name} ({status}): {message or 'no message'}")


_KNOBS = {"tier": "SZS_ROCM_TIER", "swap": "SZS_ROCM_SWAP", "packed": "SZS_ROCM_PACKED", "rune_ids": "SZS_ROCM_RUNE_IDS",
          "chain_waves": "SZS_ROCM_CHAIN_WAVES", "trace": "SZS_ROCM_TRACE", "cells": "SZS_ROCM_CELLS",
          "planner": "SZS_ROCM_PLANNER", "speculate": "SZS_ROCM_SPECULATE", "cpu_requests": "SZS_ROCM_CPU_REQUESTS",
          "streams": "SZS_ROCM_STREAMS", "reuse": "SZS_ROCM_REUSE",
          "split": "SZS_ROCM_SPLIT", "alphabet": "SZS_ROCM_ALPHABET", "merge": "SZS_ROCM_MERGE",
          "team": "SZS_ROCM_TEAM", "queues": "SZS_ROCM_QUEUES", "roctx": "SZS_ROCM_ROCTX",
          "queue": "SZS_ROCM_QUEUE", "queue_words": "SZS_ROCM_QUEUE_WORDS", "queue_rounds": "SZS_ROCM_QUEUE_ROUNDS",
          "queue_priority": "SZS_ROCM_QUEUE_PRIORITY", "fused": "SZS_ROCM_FUSED", "tiny": "SZS_ROCM_TINY"}
_knob_values = {name: os.environ.get(variable) for name, variable in _KNOBS.items()}  # what the library read when it was loaded


def tuning_set(knob: str, value=None):
    """`szs_rocm_tuning_set`: pins one of the library's tuning / testing knobs ("tier", "swap", "packed", "rune_ids",
    "chain_waves", "trace", "cells", "planner", "speculate" or the `SZS_ROCM_*` spelling); None restores the automatic
    choice.  The environment is only read once, when the library is loaded.  Returns the previous setting."""
    name = next((short for short, variable in _KNOBS.items() if knob in (short, variable)), None)
    if name is None or lib.szs_rocm_tuning_set(name.encode(), None if value is None else str(value).encode()) != 0:
        raise ValueError(f"unknown tuning knob {knob!r}")
    previous, _knob_values[name] = _knob_values[name], None if value is None else str(value)
    return previous


def team_shapes():
    """The compiled instances of the team tier: values for the `team` knob (lanes * 10000 + registers * 100 + waves)."""
    shapes, index = [], 0
    while lib.szs_rocm_team_shape(index):
        shapes.append(int(lib.szs_rocm_team_shape(index)))
        index += 1
    return shapes


def check(status: int, error: c_char_p) -> None:
    if status != 0:
        raise StringZillasError(status, error.value.decode() if error.value else None)
