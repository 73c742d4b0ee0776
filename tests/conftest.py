"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol export, gloo sharding (runs anywhere).
`-m gpu`       : parity of the HIP path against the oracle, called through the C-ABI (needs an MI355X).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding

    binding.build(with_reference=True)
    return binding.oracle()


@pytest.fixture(scope="session")
def golden():
    import json

    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "reference_matrices.json")) as f:
        matrices = json.load(f)
    with open(os.path.join(here, "known_answers.json")) as f:
        kats = json.load(f)
    return matrices, kats
