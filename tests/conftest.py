import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (hipcc cross-compiles without a GPU) and load the C-ABI library."""
    from multiscale_variational_autoencoder_amd import _build, _abi
    _build.build()
    return _abi.load_library()
