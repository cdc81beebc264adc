import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def fsamd():
    import fsamd_loader
    return fsamd_loader.load()


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    # the oracle is test infrastructure; build it once per session (gcc, seconds)
    from oracle import oracle3d
    oracle3d.build()
