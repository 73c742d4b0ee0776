"""stringzilla_amd - the MI355X-native build of StringZillas' batched similarity engines, Python side.

Mirrors the reference's `stringzillas` Python module for this hot path - same class names, constructor arguments,
call convention and result dtypes (/root/reference/python/README.md:452-560, python/stringzillas/similarities.c) -
on top of the C-ABI of `libstringzillas_rocm_shared.so`:

    import stringzilla_amd as szs
    gpu = szs.DeviceScope(gpu_device=0)
    engine = szs.LevenshteinDistances(capabilities=gpu)
    distances = engine(szs.Strs(["hello", "world"]), szs.Strs(["hallo", "word"]), device=gpu)   # 2x2 uint64

Every score is computed by hand-written gfx950 kernels behind the C-ABI.  There is no CPU path: without the shared
library, or without a GPU, construction fails loudly.  PyTorch is used only to own device memory.
"""

from __future__ import annotations

import ctypes
from typing import Iterable, Optional, Sequence, Union

import numpy as np

from . import _abi
from ._abi import CallProfile, StringZillasError, lib

__all__ = [
    "DeviceScope", "Strs", "LevenshteinDistances", "LevenshteinDistancesUTF8", "NeedlemanWunschScores",
    "SmithWatermanScores", "Fingerprints", "StringZillasError", "to_device", "Node", "NodeEngine", "__capabilities__",
    "__version__",
]

__version__ = f"{lib.szs_version_major()}.{lib.szs_version_minor()}.{lib.szs_version_patch()}"


def _capability_names(mask: int) -> tuple:
    names = []
    if mask & _abi.CAP_SERIAL:
        names.append("serial")
    if mask & _abi.CAP_CUDA:
        names.append("cuda")  # the reference's name for "a GPU engine exists"; on this build it means HIP/gfx950
    return tuple(names)


__capabilities__ = _capability_names(lib.szs_capabilities())


def _capability_mask(capabilities) -> int:
    """Accepts what the reference accepts: None, a tuple of names, or a DeviceScope (README.md:492)."""
    if capabilities is None:
        return lib.szs_capabilities()
    if isinstance(capabilities, DeviceScope):
        return capabilities.capabilities_mask
    mask = 0
    for name in capabilities:
        if name == "serial":
            mask |= _abi.CAP_SERIAL
        elif name == "parallel":
            mask |= _abi.CAP_PARALLEL
        elif name in ("cuda", "rocm", "hip"):
            mask |= _abi.CAP_CUDA
        else:
            raise ValueError(f"Unknown capability {name!r}")
    return mask


class DeviceScope:
    """`DeviceScope(cpu_cores=None, gpu_device=None)` - where engine calls run (README.md:458-470)."""

    def __init__(self, cpu_cores: Optional[int] = None, gpu_device: Optional[int] = None):
        if cpu_cores is not None and gpu_device is not None:
            raise ValueError("Cannot specify both cpu_cores and gpu_device")
        handle, error = ctypes.c_void_p(), ctypes.c_char_p()
        if gpu_device is not None:
            status = lib.szs_device_scope_init_gpu_device(gpu_device, ctypes.byref(handle), ctypes.byref(error))
        elif cpu_cores is not None:
            status = lib.szs_device_scope_init_cpu_cores(cpu_cores, ctypes.byref(handle), ctypes.byref(error))
        else:
            status = lib.szs_device_scope_init_default(ctypes.byref(handle), ctypes.byref(error))
        _abi.check(status, error)
        self.handle = handle
        self.gpu_device = gpu_device
        self.cpu_cores = cpu_cores

    @property
    def capabilities_mask(self) -> int:
        mask, error = ctypes.c_int(), ctypes.c_char_p()
        _abi.check(lib.szs_device_scope_get_capabilities(self.handle, ctypes.byref(mask), ctypes.byref(error)), error)
        return mask.value

    @property
    def capabilities(self) -> tuple:
        return _capability_names(self.capabilities_mask)

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle and lib is not None:  # `lib` is gone when the interpreter is tearing the module down
            lib.szs_device_scope_free(handle)
            self.handle = None


_default_scope: Optional[DeviceScope] = None


def _get_default_scope() -> DeviceScope:
    global _default_scope
    if _default_scope is None:
        _default_scope = DeviceScope()
    return _default_scope


class Strs:
    """An Arrow-like tape of byte strings - the stand-in for the reference's `stringzilla.Strs` (a tape is what the
    binding hands to `szs_*_u32tape` / `_u64tape`: python/stringzillas/similarities.c:275-318).

    The tape is assembled on the host and moved to device memory on first use (the reference's binding swaps the
    collection to its unified allocator at the same point: similarities.c:268-272).  A tape that was BORN on a device
    (`from_device`: the receiving side of an RCCL broadcast) keeps its bytes there; only the offsets - which the host
    planner and the row dealer read - are mirrored on the host, the bytes are downloaded if somebody asks for them."""

    def __init__(self, strings: Iterable[Union[str, bytes, bytearray, memoryview]] = (), wide_offsets: bool = False):
        encoded = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
        total = sum(len(s) for s in encoded)
        self.wide_offsets = bool(wide_offsets or total >= 2**32)
        self.count = len(encoded)
        offsets = np.zeros(self.count + 1, dtype=np.uint64 if self.wide_offsets else np.uint32)
        if self.count:
            np.cumsum(np.fromiter((len(s) for s in encoded), dtype=np.uint64, count=self.count), out=offsets[1:])
        self.offsets = offsets
        self._data = np.frombuffer(b"".join(encoded), dtype=np.uint8).copy() if total else np.zeros(1, np.uint8)
        self._device = None  # (device index or "cpu", data tensor, offsets tensor)

    @classmethod
    def from_tape(cls, data: np.ndarray, offsets: np.ndarray) -> "Strs":
        self = cls.__new__(cls)
        self._data = np.ascontiguousarray(data, dtype=np.uint8) if data.size else np.zeros(1, np.uint8)
        self.offsets = np.ascontiguousarray(offsets)
        assert self.offsets.dtype in (np.uint32, np.uint64)
        self.wide_offsets = self.offsets.dtype == np.uint64
        self.count = len(offsets) - 1
        self._device = None
        return self

    @classmethod
    def from_device(cls, data, offsets) -> "Strs":
        """A tape whose bytes (`uint8` tensor) and offsets (`int32` / `int64` tensor holding the unsigned values) already
        live in torch memory - device memory after an RCCL broadcast, host memory under `gloo`."""
        self = cls.__new__(cls)
        self._data = None
        self.wide_offsets = offsets.element_size() == 8
        self.offsets = offsets.cpu().numpy().view(np.uint64 if self.wide_offsets else np.uint32)
        self.count = len(self.offsets) - 1
        self._device = (data.device.index if data.device.type == "cuda" else "cpu", data, offsets)
        return self

    @property
    def data(self) -> np.ndarray:
        if self._data is None:  # born on a device: download on demand
            host = self._device[1].cpu().numpy()
            self._data = host if host.size else np.zeros(1, np.uint8)
        return self._data

    def __len__(self) -> int:
        return self.count

    def __getitem__(self, index: int) -> bytes:
        return self.data[int(self.offsets[index]):int(self.offsets[index + 1])].tobytes()

    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets.astype(np.int64))

    def select(self, rows: np.ndarray) -> "Strs":
        """Sub-tape holding `rows` of this tape, in that order; gathered where the bytes live (on the device for a
        device-born tape: one index computation and one gather, nothing crosses the host link)."""
        rows = np.asarray(rows, dtype=np.int64)
        offsets = self.offsets.astype(np.int64)
        starts, lengths = offsets[rows], offsets[rows + 1] - offsets[rows]
        new_offsets = np.zeros(len(rows) + 1, dtype=self.offsets.dtype)
        np.cumsum(lengths, out=new_offsets[1:])
        total = int(new_offsets[-1])
        if self._data is None and self._device is not None:
            import torch

            _, data, _ = self._device
            if total:
                shift = torch.from_numpy(np.repeat(starts - new_offsets[:-1].astype(np.int64), lengths)).to(data.device)
                gathered = data[torch.arange(total, device=data.device) + shift]
            else:
                gathered = torch.zeros(1, dtype=torch.uint8, device=data.device)
            signed = torch.from_numpy(new_offsets.view(np.int64 if self.wide_offsets else np.int32)).to(data.device)
            return Strs.from_device(gathered, signed)
        if total:
            index = np.arange(total, dtype=np.int64) + np.repeat(starts - new_offsets[:-1].astype(np.int64), lengths)
            return Strs.from_tape(self.data[index], new_offsets)
        return Strs.from_tape(np.zeros(1, np.uint8), new_offsets)

    def to_device(self, gpu_device: int = 0) -> "Strs":
        if self._device is None or self._device[0] != gpu_device:
            import torch

            if not torch.cuda.is_available():
                raise RuntimeError("stringzilla_amd needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
            where = torch.device("cuda", gpu_device)
            if self._device is not None and self._data is None:  # born elsewhere: move the tensors
                self._device = (gpu_device, self._device[1].to(where), self._device[2].to(where))
                return self
            data = torch.from_numpy(self.data).to(where)
            # torch has no uint32/uint64 arithmetic, but plain storage is all we need: view as signed of equal width.
            signed = self.offsets.view(np.int64 if self.wide_offsets else np.int32)
            offsets = torch.from_numpy(signed).to(where)
            self._device = (gpu_device, data, offsets)
        return self

    def _tape(self, gpu_device: int):
        self.to_device(gpu_device)
        _, data, offsets = self._device
        tape_type = _abi.U64Tape if self.wide_offsets else _abi.U32Tape
        return tape_type(data.data_ptr(), offsets.data_ptr(), self.count)


def to_device(strs: Strs, gpu_device: int = 0) -> Strs:
    """Module-level `szs.to_device(strs)` of the reference (README.md:560): forces device residency now."""
    return strs.to_device(gpu_device)


def _as_strs(collection) -> Strs:
    return collection if isinstance(collection, Strs) else Strs(collection)


class _Engine:
    """Shared call path: `engine(queries, candidates=None, device=None, out=None)` (README.md:472-482)."""

    _free = None
    _call_u32 = None
    _call_u64 = None
    _dtype = np.uint64

    def __init__(self):
        self.handle = ctypes.c_void_p()
        self._scope: Optional[DeviceScope] = None
        self._mask = 0

    def _remember(self, capabilities):
        self._mask = _capability_mask(capabilities)
        if isinstance(capabilities, DeviceScope):
            self._scope = capabilities

    @property
    def __capabilities__(self) -> tuple:
        return _capability_names(self._mask & lib.szs_capabilities())

    def last_call_profile(self) -> CallProfile:
        profile = CallProfile()
        if lib.szs_rocm_last_call_profile(self.handle, ctypes.byref(profile)) != 0:
            raise RuntimeError("engine is not initialized")
        return profile

    def __call__(self, queries, candidates=None, device: Optional[DeviceScope] = None, out=None):
        import torch

        scope = device or self._scope or _get_default_scope()
        gpu_device = scope.gpu_device if scope.gpu_device is not None else 0
        queries = _as_strs(queries)
        candidates = None if candidates is None else _as_strs(candidates)
        if queries.wide_offsets != (candidates.wide_offsets if candidates is not None else queries.wide_offsets):
            # both sides must use one tape flavour per call, like the binding's detection order (similarities.c:275-318)
            queries = Strs.from_tape(queries.data, queries.offsets.astype(np.uint64))
            candidates = Strs.from_tape(candidates.data, candidates.offsets.astype(np.uint64))
        rows = len(queries)
        columns = rows if candidates is None else len(candidates)

        error = ctypes.c_char_p()
        call = self._call_u64 if queries.wide_offsets else self._call_u32
        torch_dtype = torch.int64  # 8-byte cells; uint64 results are reinterpreted on the way out
        if out is None:
            results = torch.empty((rows, max(columns, 1)), dtype=torch_dtype, device=torch.device("cuda", gpu_device))
            results = results[:, :columns]
            pointer, stride = results.data_ptr(), max(columns, 1)
        elif isinstance(out, np.ndarray):
            if out.shape != (rows, columns) or out.dtype.itemsize != 8 or out.strides[1] != 8:
                raise ValueError("`out` must be a (rows, columns) matrix of 8-byte cells with contiguous rows")
            results, pointer, stride = out, out.ctypes.data, out.strides[0] // 8
        else:  # a torch tensor, host or device
            if tuple(out.shape) != (rows, columns) or out.element_size() != 8 or (columns and out.stride(1) != 1):
                raise ValueError("`out` must be a (rows, columns) matrix of 8-byte cells with contiguous rows")
            if out.is_cuda:  # whatever torch still has in flight for this matrix (a fill, say) runs on TORCH's stream; the scoring
                # launches run on the scope's own: drain the former, or a late fill overwrites cells the call has already scored
                torch.cuda.current_stream(out.device).synchronize()
            results, pointer, stride = out, out.data_ptr(), out.stride(0) if rows > 1 else max(columns, 1)

        q_tape = queries._tape(gpu_device)
        c_tape = None if candidates is None else candidates._tape(gpu_device)
        status = call(self.handle, scope.handle, ctypes.byref(q_tape), None if c_tape is None else ctypes.byref(c_tape),
                      pointer, stride, ctypes.byref(error))
        _abi.check(status, error)
        if out is not None:
            return out
        return results.cpu().numpy().view(self._dtype)

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle and self._free is not None:
            self._free(handle)
            self.handle = None


class LevenshteinDistances(_Engine):
    """`LevenshteinDistances(match=0, mismatch=1, open=1, extend=1, capabilities=None)` -> uint64 matrix."""

    _free = staticmethod(lib.szs_levenshtein_distances_free)
    _call_u32 = staticmethod(lib.szs_levenshtein_distances_u32tape)
    _call_u64 = staticmethod(lib.szs_levenshtein_distances_u64tape)
    _init = staticmethod(lib.szs_levenshtein_distances_init)
    _dtype = np.uint64

    def __init__(self, match: int = 0, mismatch: int = 1, open: int = 1, extend: int = 1, capabilities=None):
        super().__init__()
        self._remember(capabilities)
        error = ctypes.c_char_p()
        status = self._init(match, mismatch, open, extend, None, self._mask, ctypes.byref(self.handle), ctypes.byref(error))
        _abi.check(status, error)


class LevenshteinDistancesUTF8(LevenshteinDistances):
    """Codepoint-level distances (`szs_levenshtein_distances_utf8*`)."""

    _free = staticmethod(lib.szs_levenshtein_distances_utf8_free)
    _call_u32 = staticmethod(lib.szs_levenshtein_distances_utf8_u32tape)
    _call_u64 = staticmethod(lib.szs_levenshtein_distances_utf8_u64tape)
    _init = staticmethod(lib.szs_levenshtein_distances_utf8_init)


class NeedlemanWunschScores(_Engine):
    """`NeedlemanWunschScores(byte_to_class, class_substitution_costs, open=-1, extend=-1, capabilities=None)`
    -> int64 matrix of global alignment scores (README.md:500-515)."""

    _free = staticmethod(lib.szs_needleman_wunsch_scores_free)
    _call_u32 = staticmethod(lib.szs_needleman_wunsch_scores_u32tape)
    _call_u64 = staticmethod(lib.szs_needleman_wunsch_scores_u64tape)
    _init = staticmethod(lib.szs_needleman_wunsch_scores_init)
    _dtype = np.int64

    def __init__(self, byte_to_class, class_substitution_costs, open: int = -1, extend: int = -1, capabilities=None):
        super().__init__()
        self._remember(capabilities)
        byte_to_class = np.ascontiguousarray(byte_to_class, dtype=np.uint8)
        costs = np.ascontiguousarray(class_substitution_costs, dtype=np.int8)
        if byte_to_class.size != 256 or costs.size != 32 * 32:
            raise ValueError("byte_to_class must hold 256 bytes and class_substitution_costs 32x32 int8 values")
        error = ctypes.c_char_p()
        status = self._init(byte_to_class.ctypes.data, costs.ctypes.data, open, extend, None, self._mask,
                            ctypes.byref(self.handle), ctypes.byref(error))
        _abi.check(status, error)


class SmithWatermanScores(NeedlemanWunschScores):
    """Same arguments, local alignment scores (`szs_smith_waterman_scores*`)."""

    _free = staticmethod(lib.szs_smith_waterman_scores_free)
    _call_u32 = staticmethod(lib.szs_smith_waterman_scores_u32tape)
    _call_u64 = staticmethod(lib.szs_smith_waterman_scores_u64tape)
    _init = staticmethod(lib.szs_smith_waterman_scores_init)


class Fingerprints:
    """`Fingerprints(ndim, window_widths=None, alphabet_size=256, seed=0, capabilities=None)` - rolling MinHash /
    Count-Min sketches (`szs_fingerprints_*`; /root/reference/python/README.md:549-590).  Called as
    `engine(texts, device=None, out=None)`, returns `(hashes, counts)`: two `uint32` matrices of shape `(len(texts), ndim)`."""

    def __init__(self, ndim: int, window_widths=None, alphabet_size: int = 256, seed: int = 0, capabilities=None):
        self.handle = ctypes.c_void_p()
        self.ndim = int(ndim)
        self._scope = capabilities if isinstance(capabilities, DeviceScope) else None
        self._mask = _capability_mask(capabilities)
        widths = None if window_widths is None else np.ascontiguousarray(window_widths, dtype=np.uint64)
        error = ctypes.c_char_p()
        status = lib.szs_fingerprints_init(self.ndim, alphabet_size, None if widths is None else widths.ctypes.data,
                                           0 if widths is None else len(widths), seed, None, self._mask,
                                           ctypes.byref(self.handle), ctypes.byref(error))
        _abi.check(status, error)

    @property
    def capabilities(self) -> tuple:  # the reference spells this one without underscores (README.md:571)
        return _capability_names(self._mask & lib.szs_capabilities())

    def __call__(self, texts, device: Optional[DeviceScope] = None, out=None):
        import torch

        scope = device or self._scope or _get_default_scope()
        gpu_device = scope.gpu_device if scope.gpu_device is not None else 0
        texts = _as_strs(texts)
        rows = len(texts)
        if out is not None:
            hashes, counts = out
            for matrix in (hashes, counts):
                if not isinstance(matrix, np.ndarray) or matrix.shape != (rows, self.ndim) or matrix.dtype != np.uint32 or (
                        rows and matrix.strides[1] != 4):
                    raise ValueError("`out` must be a pair of (len(texts), ndim) uint32 NumPy matrices with contiguous rows")
            pointers = (hashes.ctypes.data, hashes.strides[0] if rows else self.ndim * 4,
                        counts.ctypes.data, counts.strides[0] if rows else self.ndim * 4)
            device_out = None
        else:
            device_out = torch.empty((2, max(rows, 1), self.ndim), dtype=torch.int32, device=torch.device("cuda", gpu_device))
            pointers = (device_out[0].data_ptr(), self.ndim * 4, device_out[1].data_ptr(), self.ndim * 4)
        tape = texts._tape(gpu_device)
        call = lib.szs_fingerprints_u64tape if texts.wide_offsets else lib.szs_fingerprints_u32tape
        error = ctypes.c_char_p()
        _abi.check(call(self.handle, scope.handle, ctypes.byref(tape), pointers[0], pointers[1], pointers[2], pointers[3],
                        ctypes.byref(error)), error)
        if out is not None:
            return out
        both = device_out[:, :rows].cpu().numpy().view(np.uint32)
        return both[0], both[1]

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle and lib is not None:
            lib.szs_fingerprints_free(handle)
            self.handle = None


class NodeEngine:
    """One cost model on every GPU of a `Node`; `engine(queries, candidates=None, out=None)` scores the whole cross-product
    with its query rows dealt over the GPUs (`szs_rocm_node_scores_*`, csrc/host/node.c) and returns the per-GPU statistics
    of the call as a dict (the matrix lands in `out`, or is returned when `out` is None)."""

    def __init__(self, node: "Node", kind: str, *arguments):
        self.node = node
        self.handle = ctypes.c_void_p()
        error = ctypes.c_char_p()
        init = getattr(lib, f"szs_rocm_node_{kind}_init")
        if kind in ("needleman_wunsch_scores", "smith_waterman_scores"):
            byte_to_class = np.ascontiguousarray(arguments[0], dtype=np.uint8)
            costs = np.ascontiguousarray(arguments[1], dtype=np.int8)
            if byte_to_class.size != 256 or costs.size != 32 * 32:
                raise ValueError("byte_to_class must hold 256 bytes and class_substitution_costs 32x32 int8 values")
            status = init(node.handle, byte_to_class.ctypes.data, costs.ctypes.data, arguments[2], arguments[3],
                          ctypes.byref(self.handle), ctypes.byref(error))
        else:
            status = init(node.handle, *arguments, ctypes.byref(self.handle), ctypes.byref(error))
        _abi.check(status, error)
        self._signed = kind in ("needleman_wunsch_scores", "smith_waterman_scores")
        self.last_stats: Optional[dict] = None

    def __call__(self, queries, candidates=None, out=None):
        import torch

        queries = _as_strs(queries)
        candidates = None if candidates is None else _as_strs(candidates)
        if candidates is not None and queries.wide_offsets != candidates.wide_offsets:
            queries = Strs.from_tape(queries.data, queries.offsets.astype(np.uint64))
            candidates = Strs.from_tape(candidates.data, candidates.offsets.astype(np.uint64))
        rows, columns = len(queries), len(queries if candidates is None else candidates)

        def tape_of(strs):  # where the bytes already are: a GPU if they were moved there, else host memory
            if strs._device is not None and strs._device[0] != "cpu":
                return strs._tape(strs._device[0])
            tape_type = _abi.U64Tape if strs.wide_offsets else _abi.U32Tape
            return tape_type(strs.data.ctypes.data, strs.offsets.ctypes.data, strs.count)

        q_tape = tape_of(queries)
        c_tape = None if candidates is None else tape_of(candidates)
        if out is None:
            results = np.zeros((rows, max(columns, 1)), dtype=np.int64)[:, :columns]
            pointer, stride = results.ctypes.data, max(columns, 1)
        elif isinstance(out, np.ndarray):
            if out.shape != (rows, columns) or out.dtype.itemsize != 8 or (columns and out.strides[1] != 8):
                raise ValueError("`out` must be a (rows, columns) matrix of 8-byte cells with contiguous rows")
            results, pointer, stride = out, out.ctypes.data, out.strides[0] // 8
        else:
            if tuple(out.shape) != (rows, columns) or out.element_size() != 8 or (columns and out.stride(1) != 1):
                raise ValueError("`out` must be a (rows, columns) matrix of 8-byte cells with contiguous rows")
            results, pointer, stride = out, out.data_ptr(), out.stride(0) if rows > 1 else max(columns, 1)
        stats, error = _abi.NodeStats(), ctypes.c_char_p()
        call = lib.szs_rocm_node_scores_u64tape if queries.wide_offsets else lib.szs_rocm_node_scores_u32tape
        status = call(self.handle, ctypes.byref(q_tape), None if c_tape is None else ctypes.byref(c_tape), pointer, stride,
                      ctypes.byref(stats), ctypes.byref(error))
        _abi.check(status, error)
        gpus = int(stats.gpus)
        self.last_stats = {
            "gpus": gpus, "wall_ms": float(stats.wall_milliseconds), "busy_ms": [float(stats.busy_milliseconds[i]) for i in range(gpus)],
            "kernel_ms": [float(stats.kernel_milliseconds[i]) for i in range(gpus)], "cells": [int(stats.cells[i]) for i in range(gpus)],
            "rows": [int(stats.rows[i]) for i in range(gpus)], "row_weights": [int(stats.row_weights[i]) for i in range(gpus)],
            "peer_copies": [int(stats.peer_copies[i]) for i in range(gpus)], "staged_copies": [int(stats.staged_copies[i]) for i in range(gpus)],
            "peer_pairs": int(stats.peer_pairs), "symmetric": bool(stats.symmetric),
        }
        if out is None:
            return results.view(np.int64 if self._signed else np.uint64)
        return self.last_stats

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle and lib is not None:
            lib.szs_rocm_node_engine_free(handle)
            self.handle = None


class Node:
    """`Node(gpu_devices=None)` - a set of GPUs of this host driven from ONE process, one host thread per GPU inside the
    library (`szs_rocm_node_*`).  The multi-process flavour over `torch.distributed` is `stringzilla_amd.sharded`."""

    def __init__(self, gpu_devices: Optional[Sequence[int]] = None):
        self.handle = ctypes.c_void_p()
        error = ctypes.c_char_p()
        devices = None if gpu_devices is None else np.ascontiguousarray(list(gpu_devices), dtype=np.uint64)
        status = lib.szs_rocm_node_init(None if devices is None else devices.ctypes.data, 0 if devices is None else len(devices),
                                        ctypes.byref(self.handle), ctypes.byref(error))
        _abi.check(status, error)

    def __len__(self) -> int:
        return int(lib.szs_rocm_node_size(self.handle))

    def levenshtein_distances(self, match=0, mismatch=1, open=1, extend=1) -> NodeEngine:
        return NodeEngine(self, "levenshtein_distances", match, mismatch, open, extend)

    def levenshtein_distances_utf8(self, match=0, mismatch=1, open=1, extend=1) -> NodeEngine:
        return NodeEngine(self, "levenshtein_distances_utf8", match, mismatch, open, extend)

    def needleman_wunsch_scores(self, byte_to_class, class_substitution_costs, open=-1, extend=-1) -> NodeEngine:
        return NodeEngine(self, "needleman_wunsch_scores", byte_to_class, class_substitution_costs, open, extend)

    def smith_waterman_scores(self, byte_to_class, class_substitution_costs, open=-1, extend=-1) -> NodeEngine:
        return NodeEngine(self, "smith_waterman_scores", byte_to_class, class_substitution_costs, open, extend)

    def engine_for(self, load) -> NodeEngine:
        """The engine a `workloads.Workload` names."""
        from . import matrices

        if load.kind == "levenshtein":
            return self.levenshtein_distances(**load.costs)
        if load.kind == "levenshtein_utf8":
            return self.levenshtein_distances_utf8(**load.costs)
        make = self.needleman_wunsch_scores if load.kind == "needleman_wunsch" else self.smith_waterman_scores
        return make(*matrices.by_name(load.table), **load.costs)

    def __del__(self):
        handle = getattr(self, "handle", None)
        if handle and lib is not None:
            lib.szs_rocm_node_free(handle)
            self.handle = None
