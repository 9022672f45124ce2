import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """the shared libraries are build artefacts (git-ignored): build them when a checkout does not have them yet
    (make is incremental; hipcc cross-compiles gfx950 without a GPU)"""
    need = [os.path.join(ROOT, "hemocell_amd", "lib", "libhemocell_amd.so"), os.path.join(ROOT, "oracle", "libhemo_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    from oracle import oracle as O
    return O.load()


@pytest.fixture(scope="session")
def gpu():
    """the product library on cuda:0; fails loudly if the HIP extension is missing"""
    from hemocell_amd import host
    host.init(0)
    return host
