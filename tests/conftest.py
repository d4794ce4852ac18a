import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """make sure the product library and the oracle are built"""
    import __graft_entry__ as G
    G.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    import eventql_amd as E
    c = E.Context(0)
    yield c
    c.close()
