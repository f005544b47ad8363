import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WMAP = os.path.join(GOLDEN, "wmap1new.pow")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (test infrastructure only)"""
    from oracle import zdo
    zdo.build()
    return zdo


@pytest.fixture(scope="session")
def wmap_path():
    return WMAP
