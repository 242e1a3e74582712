"""No GPU: the pieces of bench.py that decide what a record may contain (VERDICT r05 item 4b and the parity sub-record)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_refuses_child_processes_under_a_profiler(monkeypatch):
    """Under rocprofv3 the profiler's preloaded library has initialised the GPU before bench.py starts: the cold jobs (child processes) must not run then."""
    bench = importlib.import_module("bench")
    for k in list(os.environ):
        if k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    assert not bench.profiler_attached()
    monkeypatch.setenv("ROCPROFILER_REGISTER_FORCE_LOAD", "1")
    assert bench.profiler_attached()
    monkeypatch.delenv("ROCPROFILER_REGISTER_FORCE_LOAD")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.profiler_attached()


def test_film_crc_is_a_function_of_the_crop_bits_only():
    bench = importlib.import_module("bench")
    rng = np.random.default_rng(1)
    film = rng.random((1084, 1924, 4), dtype=np.float32)
    a = bench.film_crc(film, 2)
    other = film.copy(); other[0, 0, 0] += 1.0                             # outside the crop: the same crc
    assert bench.film_crc(other, 2) == a and len(a) == 8
    other[508 + 2 + 10, 928 + 2 + 10, 1] = np.nextafter(other[508 + 2 + 10, 928 + 2 + 10, 1], np.float32(2.0))      # one bit inside the crop: another crc
    assert bench.film_crc(other, 2) != a
    assert bench.PARITY_CRC_N1 is not None and bench.PARITY_SPP_TABLE == bench.SPP


def test_profile_facts_are_stamped_with_the_live_sources():
    """profiles/pmc_latest.json describes ONE build: bench.py quotes its counter facts only when the kernel / ABI sources hash to the stamp (the last thing a round does
    to those sources is the profile run)."""
    import json
    import pytest
    bench = importlib.import_module("bench")
    facts = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    if facts["source_sha16"] != bench.source_hash():                       # (not a failure of the product: bench.py then withholds the counter facts and says why)
        pytest.skip("kernel or ABI sources changed after the profile was taken: run scripts/profile_bench.sh + scripts/summarize_profile.py")
    assert os.path.isdir(os.path.join(ROOT, facts["profile"])) and "kz_wf_trace<0>" in facts["kernels"]
