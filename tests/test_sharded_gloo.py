"""The N > 1 path on CPU: world_size-2 `gloo`, two real processes.  The scorer is a stand-in (the oracle - allowed in
tests only) so that the row partition, the tape broadcast and the result reassembly run without a GPU."""
import os
import socket
import sys

import numpy as np
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
    score = lambda queries, candidates: oracle.levenshtein(strings(queries), strings(candidates))
    engine = sharded.ShardedEngine(score=score)

    load = workloads.config(5, scale=1 / 64) if rank == 0 else None  # ragged Zipf lengths; only rank 0 holds the inputs
    rows, local = engine(load.queries if load else None, load.candidates if load else None, source=0)
    full = engine(load.queries if load else None, load.candidates if load else None, source=0, gather=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rows=rows, local=local, full=full, balance=engine.last_balance)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_reassembles_the_matrix(tmp_path, oracle):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

    sys.path.insert(0, ROOT)
    from stringzilla_amd import workloads

    load = workloads.config(5, scale=1 / 64)
    strings = lambda tape: [tape[i] for i in range(len(tape))]
    expected = oracle.levenshtein(strings(load.queries), strings(load.candidates))
    seen = np.zeros(len(load.queries), dtype=bool)
    for rank in range(world):
        shard = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(shard["full"], expected)                   # gathered matrix, original row order
        assert np.array_equal(shard["local"], expected[shard["rows"]])   # this rank's rows only
        assert not seen[shard["rows"]].any()
        seen[shard["rows"]] = True
        assert 1.0 <= float(shard["balance"]) < 1.2                       # LPT keeps Zipf rows balanced
    assert seen.all()                                                     # every row scored exactly once
