import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build_oracle()
    return O


@pytest.fixture(scope="session")
def ctx_small():
    """A native context sized for the small test shapes (GPU tests only)."""
    from openvo_amd import _native
    c = _native.Context(0, 704, 512, 64, 1000)
    yield c
    c.close()
