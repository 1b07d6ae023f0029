import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        # the GPU box has 256 logical CPUs: torch's default (one thread per logical CPU) makes the CPU oracle several times SLOWER than ~one
        # thread per 8 (bench.py's sweep finds 32 fastest there); the oracle-backed tests were 400 s of the suite's 600 (round 5)
        torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8) // 4)))
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file

    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = load_file(os.path.join(REPO, "tests", "golden", name + ".safetensors"))
        return cache[name]

    return _load
