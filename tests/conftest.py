import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WMAP = os.path.join(GOLDEN, "wmap1new.pow")


def source_sha():
    """sha-256 over the native sources of the product, as zeldovich_plt_amd/csrc/Makefile records it in <library>.srcsha when it
    links a library (`make srcsha` prints the same)"""
    import hashlib
    import glob
    csrc = os.path.join(ROOT, "zeldovich_plt_amd", "csrc")
    names = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))
                   + [os.path.join(ROOT, "include", "zeldovich_hip.h")])
    h = hashlib.sha256()
    for n in names:
        h.update(open(n, "rb").read())
    return h.hexdigest()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a redundant link of a chain whose other links run every time (round 5: the GPU suite must stay "
                                       "well inside the driver's 900 s); run with ZD_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("ZD_RUN_SLOW"):
        return
    skip = pytest.mark.skip(reason="slow: a redundant link (ZD_RUN_SLOW=1 runs it)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (test infrastructure only)"""
    from oracle import zdo
    zdo.build()
    return zdo


@pytest.fixture(scope="session")
def wmap_path():
    return WMAP


class OneGroupApi:
    """zeldovich_plt_amd.api for the multi-rank parity tests written before pass groups existed: `ngpu > 1` means ONE group of
    ranks there (rows -> exchange -> planes), so make_params pins pass_groups = 1 unless a test chooses otherwise.  (The
    library's automatic choice — one GPU per group while there are enough passes — is tested in test_gpu_multi_rehearsal.py.)"""

    def __init__(self, api):
        self._api = api

    def __getattr__(self, name):
        return getattr(self._api, name)

    def make_params(self, *a, **kw):
        if kw.get("ngpu", 0) > 1:
            kw.setdefault("pass_groups", 1)
        return self._api.make_params(*a, **kw)
