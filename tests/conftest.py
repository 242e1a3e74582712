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


def wait_for_wipe(lib, limit_s=30.0):
    """Memory a process releases is wiped by the driver (~33 GB/s) before it is reported free again, and whoever allocates before the wipe is through waits for it
    (profiles/r05a_alloc). A test whose point is the TIMING of a growing context waits here until the free memory has stopped rising."""
    import ctypes as C
    import time
    f, t = C.c_uint64(), C.c_uint64()
    last, still, t0 = -1, 0, time.perf_counter()
    while time.perf_counter() - t0 < limit_s:
        assert lib.kz_device_mem_info(0, C.byref(f), C.byref(t)) == 0
        still = still + 1 if f.value <= last + (64 << 20) else 0
        last = max(last, f.value)
        if still >= 4:
            break
        time.sleep(0.1)
    return last


@pytest.fixture(scope="session")
def dev_lib(kz, gpu_lib):
    """The development variant of the library (-DKZ_EXPERIMENTS: same sources + the hooks that are process-global state - failure injection, growth delay,
    trace, device aliasing - and the kernels of rejected experiments). A second copy of the library in this process: its replicas, pools and device state are its
    own. The product library is asked to give its pooled path state back first (the two copies share one card)."""
    if not os.path.exists(kz.abi.DEV_LIB_PATH):
        pytest.fail("development variant not built: python -c 'import __graft_entry__ as g; g.build()'")
    gpu_lib.kz_device_trim(0)
    wait_for_wipe(gpu_lib)
    lib = kz.abi.load_dev_library()
    assert lib.kz_build_flags() & 1 and lib.kz_device_count() >= 1
    return lib
