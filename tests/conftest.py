import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "deep-neuroevolution_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def small_noise():
    """First 4M entries of the reference noise stream (es.py:60, seed 123)."""
    return np.random.RandomState(123).randn(4_000_000).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O
