import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """build the oracle and the in-tree libraries if they are not there yet (hipcc cross-compiles without a GPU)"""
    import poroelasticity_dealii_amd as pk
    if not (os.path.exists(os.path.join(pk.LIB_DIR, "libporoel_hip.so")) and os.path.exists(os.path.join(pk.LIB_DIR, "libporoel_host.so"))):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    import oracle_py
    oracle_py.load()
