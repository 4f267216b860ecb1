import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, HERE):     # the repo root (product, oracle) and tests/ itself (gpu_common, rng_spec)
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(scope="session")
def ra():
    """The product package with the device initialised (GPU tests)."""
    import riemann_amd
    from riemann_amd import _lib
    _lib.call("rmh_init", 0)
    return riemann_amd


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    return oracle
