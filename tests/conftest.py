import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle is built on demand (gcc, seconds); libramx.so must already be there
    (python -c 'import __graft_entry__ as g; g.build()') -- tests never build the product silently."""
    from oracle import pyoracle as po
    if not os.path.exists(os.path.join(ROOT, "oracle", "libramx_oracle.so")):
        po.build(ref=False)
    yield
