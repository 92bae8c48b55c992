import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "progressive-x_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pgx_oracle
    pgx_oracle.lib()
    return pgx_oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One libpgx context on device 0 for the whole session; fails loudly if the HIP library or the GPU is missing."""
    from pyprogressivex import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
