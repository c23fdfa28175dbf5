import os
import sys

import pytest

# the tests switch between kernel generations through the library's LGU_* debug variables: those are honoured only when
# this is set BEFORE the library is loaded (include/lgu_corr.h; tests/test_abi.py checks the gate itself)
os.environ.setdefault("LGU_DEBUG_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def lgu():
    import lgu_slam_amd
    return lgu_slam_amd
