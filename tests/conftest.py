import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is git-ignored: (cross-)compile it on first use so a fresh checkout can run the suite
    from apr_amd import build as _b
    if _b.needs_build():
        _b.build(verbose=False)


@pytest.fixture(scope="session")
def lib():
    from apr_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
