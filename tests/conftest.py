import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product package (ctypes binding over librtamd.so / librtamd_host.so)."""
    return entry.load_package()


@pytest.fixture(scope="session")
def ol():
    """The CPU oracle binding (test infrastructure)."""
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib


@pytest.fixture(scope="session")
def sky(ol):
    return ol.procedural_sky()
