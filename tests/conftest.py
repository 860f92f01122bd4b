import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout(seconds): pytest-timeout limit (a no-op marker when the plugin is absent)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/, test infrastructure)."""
    import oracle as _orc
    _orc.build()
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def alg():
    """The host package (product)."""
    import algames_jl_amd
    return algames_jl_amd
