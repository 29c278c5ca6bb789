import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """A fresh checkout has no built library (*.so is git-ignored): build it once, as
    __graft_entry__.build() does -- hipcc cross-compiles without a GPU.  This is not a
    fallback: the tests still go through libeggshell_amd.so and nothing else."""
    lib = os.path.join(ROOT, "eggshell_amd", "libeggshell_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def ctx():
    """One egs_context per test session (GPU tests only)."""
    from eggshell_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
