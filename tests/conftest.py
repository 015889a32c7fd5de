import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def refprims():
    from oracle.oracle_py import RefPrims
    if not RefPrims.available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return RefPrims()


@pytest.fixture()
def rng():
    return np.random.default_rng(20261003)
