import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def golden():
    """name -> dict of arrays from tests/golden/<name>.npz (generated from the reference)."""
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
        return cache[name]
    return load


@pytest.fixture(scope="session")
def dev():
    if not has_gpu():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
