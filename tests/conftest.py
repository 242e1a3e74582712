import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kz():
    return importlib.import_module("nano-kazen_amd")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure). Builds oracle/liboracle.so on first use."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_lib(kz):
    """The product library on a GPU box; fails loudly (no silent skip) if the HIP extension is missing."""
    lib = kz.abi.load_library()
    assert lib.kz_device_count() >= 1, "no HIP device visible: -m gpu tests must run on the GPU box"
    return lib
