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
    config.addinivalue_line("markers", "perf: wall-clock guards, run explicitly with -m perf on the GPU box (never part of -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_ptq_small.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_ptq_slices.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_qat_small.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_qlinear.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_block_small.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_kat.npz")
