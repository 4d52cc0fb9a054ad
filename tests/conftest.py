import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A kernel that never signals completion would leave the host polling for ever and the GPU box unusable: every
    device test gets a wall-clock bound (pytest-timeout's thread method ends the process, which also frees the
    device; the signal method cannot interrupt a blocked C call)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(420, method="thread"))


def pytest_collection_finish(session):
    """On a cold GPU box the first `import torch` pages the ROCm libraries in and has been seen to take minutes:
    pay that here, outside the wall-clock bound of whichever device test would import it first."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


def pytest_sessionstart(session):
    """libmsi.so is a build artefact (git-ignored): build it when a fresh checkout runs the tests before
    __graft_entry__.build() did (hipcc cross-compiles gfx950 without a GPU; `make` is a no-op when up to date)."""
    import subprocess
    so = os.path.join(ROOT, "meilisearch_amd", "libmsi.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "meilisearch_amd", "csrc"), "ARCH=gfx950"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never imported by the product)."""
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def cpubase():
    from oracle import cpubase
    cpubase.lib()
    return cpubase


@pytest.fixture(scope="session")
def ctx():
    """One libmsi context for the whole GPU session.  No skip, no fallback: if the
    HIP library or the device is missing the GPU tests must fail loudly."""
    import meilisearch_amd as ma
    c = ma.Context(0)
    yield c
    c.close()
