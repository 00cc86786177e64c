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


@pytest.fixture
def tune(monkeypatch):
    """tune(name, value): pin a library option (include/snk.h "tuning"; names as snk_option_name lists them, the old SNK_* spelling is
    accepted) for the rest of this test -- on every open Engine (snk_ctx_set_option) and, through SNK_TUNING, on the contexts created
    later (in-process ranks, subprocesses).  Undone at teardown."""
    import os
    from supernova_amd import engine as _e
    applied = []

    def _set(name, value):
        name = name[4:].lower() if name.startswith("SNK_") else name
        value = int(value, 0) if isinstance(value, str) else int(value)
        items = [i for i in os.environ.get("SNK_TUNING", "").split(",") if i and not i.startswith(name + "=")]
        monkeypatch.setenv("SNK_TUNING", ",".join(items + [f"{name}={value}"]))
        for e in _e.live_engines():
            applied.append((e, name, e.get_option(name)))
            e.set_option(name, value)

    yield _set
    for e, name, prev in reversed(applied):
        if getattr(e, "_ctx", None):
            e.set_option(name, prev) if prev is not None else e.clear_option(name)
