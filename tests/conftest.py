import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def snk():
    """libsnk.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    from supernova_amd import build, lib
    build.build(verbose=False)
    return lib.load()
