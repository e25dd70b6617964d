import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def sample_idx(numel, h, k=16):
    from ava_amd import synthetic as syn
    return np.minimum((syn.u01(k, 5000 + h) * numel).astype(np.int64), numel - 1)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
