import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


from __graft_entry__ import load_package  # noqa: E402

load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_spectral():
    import numpy as np
    return np.load(os.path.join(HERE, "golden", "spectral_golden.npz"))
