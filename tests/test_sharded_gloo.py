"""The N > 1 path on CPU: `gloo`, world sizes 2 and 8, real processes.  The scorer is a stand-in (the oracle - allowed in
tests only) so that the row partition, the tape broadcast and the result reassembly run without a GPU: rows dealt by LPT,
symmetric calls as bands of the lower triangle (each rank scores a rectangle and a triangle, together exactly the cells one
symmetric engine call scores), ranks left without a row, and the gathered matrices against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    from stringzilla_amd import sharded, workloads

    oracle = binding.oracle()
    strings = lambda tape: [tape[i] for i in range(len(tape))]
    scored = []  # cells this rank scored: sum over its engine calls of the pairs' len x len

    def score(queries, candidates):
        q = strings(queries)
        if candidates is None:  # the band's own triangle: the engine scores j <= i once and mirrors
            lengths = np.array([len(x) for x in q], dtype=np.int64)
            scored.append(int((lengths * np.cumsum(lengths)).sum()))
            return oracle.levenshtein(q, None)
        c = strings(candidates)
        scored.append(sum(map(len, q)) * sum(map(len, c)))
        return oracle.levenshtein(q, c)

    engine = sharded.ShardedEngine(score=score)

    load = workloads.config(5, scale=1 / 64) if rank == 0 else None  # ragged Zipf lengths; only rank 0 holds the inputs
    rows, local = engine(load.queries if load else None, load.candidates if load else None, source=0)
    balance = engine.last_balance
    full = engine(load.queries if load else None, load.candidates if load else None, source=0, gather=True)
    # symmetric: the source rank passes no candidates; the lower triangle is dealt in bands and scored ONCE
    del scored[:]
    band_rows, band = engine(load.queries if load else None, None, source=0)
    band_cells, band_balance = sum(scored), engine.last_balance
    mirrored = engine(load.queries if load else None, None, source=0, gather=True)
    # fewer rows than ranks: some ranks score nothing at all, in both modes
    few = workloads.config(5, scale=1 / 640) if rank == 0 else None
    few_full = engine(few.queries if few else None, few.candidates if few else None, source=0, gather=True)
    few_mirrored = engine(few.queries if few else None, None, source=0, gather=True)
    _, few_band = engine(few.queries if few else None, None, source=0)  # ranks without a row still return the scorer's KIND of matrix
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=rows, local=local, full=full, balance=balance, band_rows=band_rows,
             band=band, band_cells=band_cells, band_balance=band_balance, mirrored=mirrored, few_full=few_full, few_mirrored=few_mirrored,
             few_band_kind=f"{type(few_band).__name__}:{few_band.dtype}:{few_band.shape[0]}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharding_reassembles_the_matrix(tmp_path, oracle, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

    sys.path.insert(0, ROOT)
    from stringzilla_amd import workloads

    load, few = workloads.config(5, scale=1 / 64), workloads.config(5, scale=1 / 640)
    strings = lambda tape: [tape[i] for i in range(len(tape))]
    expected = oracle.levenshtein(strings(load.queries), strings(load.candidates))
    symmetric = oracle.levenshtein(strings(load.queries), None)
    lengths = load.queries.lengths().astype(np.int64)
    triangle_cells = int((lengths * np.cumsum(lengths)).sum())
    seen = np.zeros(len(load.queries), dtype=bool)
    band_seen = np.zeros(len(load.queries), dtype=bool)
    band_cells = 0
    for rank in range(world):
        shard = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(shard["full"], expected)                   # gathered matrix, original row order
        assert np.array_equal(shard["local"], expected[shard["rows"]])   # this rank's rows only
        assert not seen[shard["rows"]].any()
        seen[shard["rows"]] = True
        assert 1.0 <= float(shard["balance"]) < (1.2 if world == 2 else 2.0)  # LPT keeps Zipf rows balanced (49 rows over 8 ranks: coarser)
        # symmetric: the mirrored matrix on every rank; the band holds the lower triangle of its rows and nothing above it
        assert np.array_equal(shard["mirrored"], symmetric)
        rows = shard["band_rows"]
        if len(rows):
            assert np.array_equal(rows, np.arange(rows[0], rows[-1] + 1)) and not band_seen[rows].any()  # contiguous, disjoint
            band_seen[rows] = True
            end = int(rows[-1]) + 1
            assert np.array_equal(shard["band"][:, :end], symmetric[rows][:, :end]) and not shard["band"][:, end:].any()
        band_cells += int(shard["band_cells"])
        assert np.array_equal(shard["few_full"], oracle.levenshtein(strings(few.queries), strings(few.candidates)))
        assert np.array_equal(shard["few_mirrored"], oracle.levenshtein(strings(few.queries), None))
    kinds = [str(np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))["few_band_kind"]) for rank in range(world)]
    assert len({kind.rsplit(":", 1)[0] for kind in kinds}) == 1 and kinds[0].startswith("ndarray:uint64"), kinds  # one kind on every rank,
    assert world == 2 or any(kind.endswith(":0") for kind in kinds)                                              # also on those without a row
    assert seen.all() and band_seen.all()                                 # every row scored exactly once, in both modes
    assert band_cells == triangle_cells                                   # the ranks together scored the TRIANGLE, not the square
