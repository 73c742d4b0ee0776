"""`-m gpu`: bench.py's N > 1 code path on a one-GPU box - two real ranks under `torch.distributed.run`, both on cuda:0
(`--same-device --backend gloo`): the weak-scaled headline, the strong-scaled records of configs 4 and 5 (rows dealt by LPT
over the ranks, tapes broadcast and kept where the collective delivered them) and the single-process C entry
(`szs_rocm_node_*`) over the same "GPUs".  The numbers of such a run mean nothing; the checksums must equal a plain
single-GPU computation of the same (scaled) batches."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks,launcher", [(2, "torchrun"), (8, "torchrun"), (2, "self")])
def test_ranks_on_one_device(ranks, launcher, tmp_path):
    """Two ranks, and EIGHT - the world size BASELINE.json's curve ends at, never run anywhere before round 4 - under the driver's
    launcher; and plain `python bench.py --gpus 2`, which starts its own ranks (round 5: it used to assert on WORLD_SIZE)."""
    import stringzilla_amd as szs
    from stringzilla_amd import matrices, workloads

    scale = 1 / 8
    options = ["--gpus", str(ranks), "--steps", "5", "--warmup", "2", "--backend", "gloo", "--same-device", "--extra-scale", str(scale),
               "--extra-seconds", "0.2", "--cpu-seconds", "1", "--extra-cpu-seconds", "0.5",
               "--details", str(tmp_path / "bench_configs.json")]  # not gpurun_out/bench_configs.json: that is the real run's
    if launcher == "torchrun":
        command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + options
    else:
        command = [sys.executable, os.path.join(ROOT, "bench.py")] + options
    environment = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    done = subprocess.run(command, cwd=ROOT, capture_output=True, text=True, timeout=900, env=environment)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    printed = [json.loads(text) for text in done.stdout.splitlines() if text.startswith("{")]
    line = printed[-1]
    assert len(json.dumps(line)) < 4096 and "configs" not in line  # the last line is the headline alone (tests/test_bench_line.py)
    assert line["n_gpus"] == ranks and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["pairs_per_gpu"] == 1024 * 1024
    assert "alternate" in line["config"]["stream"] and line["same_tapes"]["value"] > 0  # the headline is the fresh-batch stream
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1     # reported on rank 0 whatever N
    assert line["roofline"]["frac"] > 0

    gpu = szs.DeviceScope(gpu_device=0)
    expected = {}
    for config in (2, 4, 5):
        # the batches bench.py scores: std::mt19937_64 for configs 1-4 when the helper library is built, numpy otherwise
        load = workloads.config(config, scale=scale, generator="mt19937_64" if config <= 4 else "numpy")
        if load.kind == "levenshtein":
            engine = szs.LevenshteinDistances(**load.costs, capabilities=gpu)
        else:
            engine = szs.SmithWatermanScores(*matrices.by_name(load.table), **load.costs, capabilities=gpu)
        expected[config] = (int(engine(load.queries, load.candidates, device=gpu).view(np.int64).sum()), load.cells, len(load.queries))
    records = [entry["configs_record"] for entry in printed if "configs_record" in entry]
    assert set(line["configs_gcups"]) >= {"2", "4", "5", "2@node", "4@node", "5@node"}
    assert line["ranks"] == {"backend": "gloo", "world": ranks, "devices": 1}  # what the collective layer says about the job
    strong = {record["config"]: record for record in records if record.get("scaling") == "strong" and "sharding" in record}
    node = {record["config"]: record for record in records if record.get("entry_point", "").startswith("szs_rocm_node")}
    for config in (2, 4, 5):  # 2: the metric's own batch, strong-scaled beside the weak-scaled headline (round 6)
        checksum, cells, rows = expected[config]
        assert "error" not in strong[config] and "error" not in node[config], (strong[config], node[config])
        assert strong[config]["results_checksum"] == checksum and strong[config]["cells"] == cells
        assert sum(strong[config]["rows_per_gpu"]) == rows and len(strong[config]["busy_ms_per_gpu"]) == ranks
        assert strong[config]["imbalance_max_over_mean"] >= 1.0
        assert 0 < strong[config]["roofline"]["frac"] <= 1 and strong[config]["roofline"]["peak"] == 8000.0 * ranks
        assert strong[config]["ranks"] == ranks and strong[config]["backend"] == "gloo" and strong[config]["kernel_gcups"] > 0
        assert node[config]["results_checksum"] == checksum and sum(node[config]["rows_per_gpu"]) == rows
        assert len(node[config]["busy_ms_per_gpu"]) == ranks and all(ms > 0 for ms in node[config]["busy_ms_per_gpu"])


def test_two_ranks_on_two_devices_over_rccl(tmp_path):
    """The path the driver's 8-GPU run takes - `--backend nccl` (RCCL), one rank per DEVICE - on the smallest box that has it: two
    GPUs.  Skipped on a one-GPU box (every `gpurun` box of this pool): there the same code runs over `gloo` on one device, above.
    Checks what only distinct devices can show: the broadcast tapes arrive on the other device, every rank scores on its own GPU,
    the strong-scaled checksums equal a single-GPU computation."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs two GPUs, {torch.cuda.device_count()} visible")
    import stringzilla_amd as szs
    from stringzilla_amd import workloads

    scale = 1 / 8
    command = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--backend", "nccl",
               "--extra-scale", str(scale), "--extra-seconds", "0.2", "--cpu-seconds", "1", "--extra-cpu-seconds", "0.5", "--extra-configs", "2,5",
               "--details", str(tmp_path / "bench_configs.json")]
    environment = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    environment["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"  # the host driver only supports dmabuf IPC
    done = subprocess.run(command, cwd=ROOT, capture_output=True, text=True, timeout=900, env=environment)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    printed = [json.loads(text) for text in done.stdout.splitlines() if text.startswith("{")]
    line = printed[-1]
    assert line["n_gpus"] == 2 and line["ranks"] == {"backend": "nccl", "world": 2, "devices": 2}
    gpu = szs.DeviceScope(gpu_device=0)
    records = {record["config"]: record for record in (entry["configs_record"] for entry in printed if "configs_record" in entry)
               if record.get("scaling") == "strong" and "sharding" in record}
    for config in (2, 5):
        load = workloads.config(config, scale=scale, generator="mt19937_64" if config <= 4 else "numpy")
        engine = szs.LevenshteinDistances(**load.costs, capabilities=gpu)
        checksum = int(engine(load.queries, load.candidates, device=gpu).view(np.int64).sum())
        assert records[config]["results_checksum"] == checksum and records[config]["backend"] == "nccl" and not records[config]["same_device"]


def test_the_distributed_path_over_rccl_with_a_world_of_one(tmp_path):
    """`bench.py --gpus 1 --force-distributed --backend nccl`: the N > 1 code path of the judged command - process group over RCCL, the
    candidates' broadcast, configs 2 / 4 / 5 strong-scaled through `stringzilla_amd.sharded`, the C node driver, the all-reduces and
    barriers of the headline - with a world of ONE rank: RCCL's first contact on the one-GPU boxes of this pool (the gloo tests above
    cover two and eight ranks on one device; the two-DEVICE test skips here)."""
    import stringzilla_amd as szs
    from stringzilla_amd import matrices, workloads

    scale = 1 / 8
    command = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-distributed", "--backend", "nccl", "--steps", "5", "--warmup", "2",
               "--extra-scale", str(scale), "--extra-seconds", "0.2", "--cpu-seconds", "1", "--extra-cpu-seconds", "0.5",
               "--details", str(tmp_path / "bench_configs.json")]
    environment = {key: value for key, value in os.environ.items() if key not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    environment["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    done = subprocess.run(command, cwd=ROOT, capture_output=True, text=True, timeout=900, env=environment)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    printed = [json.loads(text) for text in done.stdout.splitlines() if text.startswith("{")]
    line = printed[-1]
    assert line["ranks"] == {"backend": "nccl", "world": 1, "devices": 1} and line["n_gpus"] == 1 and line["value"] > 0 and line["audit"] == []
    gpu = szs.DeviceScope(gpu_device=0)
    records = [entry["configs_record"] for entry in printed if "configs_record" in entry]
    strong = {record["config"]: record for record in records if record.get("scaling") == "strong" and "sharding" in record}
    node = {record["config"]: record for record in records if record.get("entry_point", "").startswith("szs_rocm_node")}
    for config in (2, 4, 5):
        load = workloads.config(config, scale=scale, generator="mt19937_64" if config <= 4 else "numpy")
        if load.kind == "levenshtein":
            engine = szs.LevenshteinDistances(**load.costs, capabilities=gpu)
        else:
            engine = szs.SmithWatermanScores(*matrices.by_name(load.table), **load.costs, capabilities=gpu)
        checksum = int(engine(load.queries, load.candidates, device=gpu).view(np.int64).sum())
        assert "error" not in strong[config] and "error" not in node[config], (strong[config], node[config])
        assert strong[config]["results_checksum"] == checksum and strong[config]["backend"] == "nccl" and strong[config]["ranks"] == 1
        assert node[config]["results_checksum"] == checksum
