import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLD):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests are skipped (not failed) on a box without a GPU only when not explicitly selected
    if torch.cuda.is_available():
        return
    sel = config.getoption("-m") or ""
    if "gpu" in sel and "not gpu" not in sel:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gold():
    def load(name):
        return np.load(os.path.join(GOLD, name))
    return load
