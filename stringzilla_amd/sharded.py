"""Node-level sharding of one cross-product over N GPUs: one process per GPU, `torch.distributed` (RCCL over xGMI).

The path shards embarrassingly: cells are independent, so query ROWS are dealt to ranks and every
rank scores `its rows x all candidates` through its own single-GPU engine - the C-ABI stays one device per scope, like
the reference's (stringzillas.h:137; the reference has no multi-GPU path at all).  The only exchange steps are

  1. replicate the inputs:  broadcast of the candidates tape and of the (small) query tape from the source rank;
  2. optionally, `gather=True`: all-gather the result row blocks so that every rank holds the full matrix.

A symmetric call (`candidates=None` on the source rank) shards the LOWER TRIANGLE instead: contiguous bands of rows of equal
weight (`szs_rocm_shard_triangle`), each a rectangle plus a triangle of its own - the ranks together score what one symmetric
engine call scores, and the gathered matrix is mirrored on the collective's device.

Rows are dealt by longest-processing-time on `len(query)` (`szs_rocm_shard_rows`), so ragged batches (config 5: Zipf
lengths) stay balanced; `last_balance` reports max/mean of the per-rank loads.
"""

from __future__ import annotations

import ctypes
from typing import Callable, Optional

import numpy as np

from . import Strs, _abi


def shard_rows(lengths: np.ndarray, shards: int):
    """LPT assignment: returns (shard_of_row[uint32], loads[uint64]); weight of a row = len(query) + 1."""
    weights = np.ascontiguousarray(lengths, dtype=np.uint64) + np.uint64(1)
    shard_of_row = np.zeros(len(weights), dtype=np.uint32)
    loads = np.zeros(shards, dtype=np.uint64)
    status = _abi.lib.szs_rocm_shard_rows(weights.ctypes.data, len(weights), shards, shard_of_row.ctypes.data, loads.ctypes.data)
    if status != 0:
        raise _abi.StringZillasError(status, "szs_rocm_shard_rows failed")
    return shard_of_row, loads


def shard_triangle(lengths: np.ndarray, shards: int):
    """Contiguous bands of the lower triangle of a symmetric call, of equal weight (`szs_rocm_shard_triangle`): returns
    (band_first[shards + 1], weights[uint64]); band g = rows [band_first[g], band_first[g + 1])."""
    lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
    band_first = np.zeros(shards + 1, dtype=np.uint64)
    weights = np.zeros(shards, dtype=np.uint64)
    status = _abi.lib.szs_rocm_shard_triangle(lengths.ctypes.data, len(lengths), shards, band_first.ctypes.data, weights.ctypes.data)
    if status != 0:
        raise _abi.StringZillasError(status, "szs_rocm_shard_triangle failed")
    return band_first.astype(np.int64), weights


class ShardedEngine:
    """Wraps a single-GPU engine (`LevenshteinDistances`, `NeedlemanWunschScores`, ...) for a process group.

    `score(queries, candidates) -> ndarray[rows, columns]` defaults to the engine call on this rank's GPU; tests on a
    GPU-less box inject a stand-in so that the partition / exchange / reassembly logic runs under `gloo`.
    """

    def __init__(self, engine=None, scope=None, group=None, score: Optional[Callable] = None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.engine = engine
        self.scope = scope
        if score is None:
            if engine is None:
                raise ValueError("ShardedEngine needs an engine (there is no CPU fallback)")

            def score(queries, candidates):  # this rank's rows x all candidates (None: x themselves), results left in HBM
                import torch

                gpu = scope.gpu_device if scope is not None and scope.gpu_device is not None else torch.cuda.current_device()
                columns = len(queries) if candidates is None else len(candidates)
                out = torch.empty((len(queries), columns), dtype=torch.int64, device=torch.device("cuda", gpu))
                return engine(queries, candidates, device=scope, out=out)

        self._score = score
        self.last_balance = 1.0
        self.last_rows: Optional[np.ndarray] = None

    def _device(self):
        import torch

        backend = self._dist.get_backend(self.group)
        if backend == "nccl":  # RCCL: collectives run on device buffers
            return torch.device("cuda", self.scope.gpu_device if self.scope is not None and self.scope.gpu_device is not None else torch.cuda.current_device())
        return torch.device("cpu")

    def _broadcast_tape(self, strs: Optional[Strs], source: int) -> Strs:
        """Replicates one tape from `source`.  The receiving side keeps bytes and offsets where the collective delivered
        them - in HBM under RCCL - and mirrors only the offsets on the host (the row dealer and the planner read them)."""
        import torch

        device = self._device()
        header = torch.zeros(3, dtype=torch.int64, device=device)
        if self.rank == source:
            header = torch.tensor([strs.data.size, strs.count, int(strs.wide_offsets)], dtype=torch.int64, device=device)
        self._dist.broadcast(header, source, group=self.group)
        size, count, wide = (int(x) for x in header.tolist())
        torch_offset = torch.int64 if wide else torch.int32
        if self.rank == source:
            if device.type == "cuda":
                strs.to_device(device.index)
                _, data, offsets = strs._device
            else:
                data = torch.from_numpy(strs.data)
                offsets = torch.from_numpy(strs.offsets.view(np.int64 if wide else np.int32))
        else:
            data = torch.empty(size, dtype=torch.uint8, device=device)
            offsets = torch.empty(count + 1, dtype=torch_offset, device=device)
        self._dist.broadcast(data, source, group=self.group)
        self._dist.broadcast(offsets, source, group=self.group)
        if self.rank == source:
            return strs
        return Strs.from_device(data, offsets)

    def _agree_on_symmetry(self, no_candidates: bool, source: int) -> bool:
        """Only `source` knows whether the call is symmetric (the other ranks pass no inputs at all): one small broadcast."""
        import torch

        flag = torch.tensor([int(no_candidates)], dtype=torch.int64, device=self._device())
        self._dist.broadcast(flag, source, group=self.group)
        return bool(int(flag.item()))

    def _symmetric(self, strings: Strs, gather: bool):
        """Self-similarity: the LOWER TRIANGLE in contiguous bands of rows of equal weight - row i weighs len_i x sum_{j <= i}
        len_j - so that the N ranks together score what one symmetric engine call scores, not the full
        square (serial.hpp:3169-3182).  A band is its rows against every string before it plus the triangle of its own rows:
        two calls of this rank's engine.  Returns (rows, local) - the band's rows with columns [0, band end) filled, the rest
        zero - or, with `gather`, the full mirrored matrix on every rank."""
        import torch

        count = len(strings)
        band_first, weights = shard_triangle(strings.lengths(), self.world)
        self.last_balance = float(weights.max() / max(weights.mean(), 1.0)) if count else 1.0
        first, end = int(band_first[self.rank]), int(band_first[self.rank + 1])
        rows = np.arange(first, end)
        self.last_rows = rows
        band = strings.select(rows)
        parts = []
        if end > first:
            if first:
                parts.append(self._score(band, strings.select(np.arange(0, first))))  # the rectangle
            parts.append(self._score(band, None))                                        # the band's own triangle, mirrored inside
        # What kind of matrix this is - a device tensor or a NumPy one, signed or unsigned cells - is the SCORER's choice, and a rank
        # whose band is empty (fewer rows than ranks, very skewed strings) has no part to read it from: the ranks agree on it (one
        # small all-reduce; -1 = "no part here"), so that every rank returns the same kind and collectives on the results line up.
        kind = torch.tensor([-1 if not parts else int(any(isinstance(part, torch.Tensor) for part in parts)),
                             -1 if not parts or isinstance(parts[0], torch.Tensor) else int(np.asarray(parts[0]).dtype == np.uint64)],
                            dtype=torch.int64, device=self._device())
        self._dist.all_reduce(kind, op=self._dist.ReduceOp.MAX, group=self.group)
        as_tensor = int(kind[0]) > 0
        agreed_dtype = np.uint64 if int(kind[1]) > 0 else np.int64
        device = self._device() if as_tensor or gather else torch.device("cpu")
        local = torch.zeros((end - first, count), dtype=torch.int64, device=device if as_tensor else torch.device("cpu"))
        column = 0
        for part in parts:
            block = part if isinstance(part, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(part).view(np.int64))
            local[:, column:column + block.shape[1]] = block.to(local.device)
            column += block.shape[1]
        dtype = np.int64 if as_tensor else agreed_dtype
        if not gather:
            return rows, (local if as_tensor else local.numpy().view(dtype))
        # equal-sized (padded) bands, all-gathered; then the cells above the diagonal from the ones below
        sizes = np.diff(band_first)
        longest = int(sizes.max()) if len(sizes) else 0
        padded = torch.zeros((longest, count), dtype=torch.int64, device=device)
        padded[:end - first] = local.to(device)
        blocks = [torch.empty_like(padded) for _ in range(self.world)]
        self._dist.all_gather(blocks, padded, group=self.group)
        full = torch.zeros((count, count), dtype=torch.int64, device=device)
        for rank, block in enumerate(blocks):
            a, b = int(band_first[rank]), int(band_first[rank + 1])
            full[a:b] = block[:b - a]
        full = torch.tril(full) + torch.tril(full, -1).T
        return full if as_tensor else full.cpu().numpy().view(dtype)

    def __call__(self, queries: Optional[Strs], candidates: Optional[Strs], source: int = 0, gather: bool = False):
        """Every rank calls this; only `source` needs to pass the inputs.  Returns (row_indices, local_matrix) - this
        rank's result rows and which global rows they are - or, with `gather=True`, the full matrix on every rank
        (a tensor on the collective's device when the scorer returned tensors, else a NumPy matrix)."""
        import torch

        symmetric = self._agree_on_symmetry(candidates is None, source)
        queries = self._broadcast_tape(queries, source)
        if symmetric:
            return self._symmetric(queries, gather)
        candidates = self._broadcast_tape(candidates, source)
        shard_of_row, loads = shard_rows(queries.lengths(), self.world)
        self.last_balance = float(loads.max() / max(loads.mean(), 1.0))
        rows = np.nonzero(shard_of_row == self.rank)[0]
        self.last_rows = rows
        local = self._score(queries.select(rows), candidates) if len(rows) else np.zeros((0, len(candidates)), dtype=np.int64)
        if not gather:
            return rows, local

        # The one collective on the result side: equal-sized (padded) row blocks, all-gathered and dealt back to their
        # global rows on the collective's device - nothing is staged through host memory under RCCL.
        device = self._device()
        columns = len(candidates)
        counts = np.bincount(shard_of_row, minlength=self.world)
        longest = int(counts.max()) if len(counts) else 0
        as_tensor = isinstance(local, torch.Tensor)
        dtype = np.int64 if as_tensor else np.asarray(local).dtype
        padded = torch.zeros((longest, columns), dtype=torch.int64, device=device)
        if len(rows):
            block = local if as_tensor else torch.from_numpy(np.ascontiguousarray(local).view(np.int64))
            padded[:len(rows)] = block.to(device)
        blocks = [torch.empty_like(padded) for _ in range(self.world)]
        self._dist.all_gather(blocks, padded, group=self.group)
        full = torch.zeros((len(queries), columns), dtype=torch.int64, device=device)
        for rank, block in enumerate(blocks):
            owned = np.nonzero(shard_of_row == rank)[0]
            if len(owned):
                full[torch.from_numpy(owned).to(device)] = block[:len(owned)]
        return full if as_tensor else full.cpu().numpy().view(dtype)
