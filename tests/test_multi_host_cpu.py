"""No GPU: the in-process multi-device driver (kz_render_multi, nano-kazen_amd/csrc/kz_multi.cpp - the drop-in's DEFAULT on a multi-GPU node) with real thread
concurrency under ThreadSanitizer. tests/host_cpp/multi_tsan_test.cpp links kz_multi.cpp and kz_plan.cpp unchanged and fakes the four
device entry points their threads call; the -m gpu half of the same item is tests/test_gpu_multi.py (aliased replicas of one GPU)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nano-kazen_amd", "csrc")


def test_multi_device_driver_is_race_free_and_order_independent(tmp_path):
    exe = str(tmp_path / "multi_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                           os.path.join(ROOT, "tests", "host_cpp", "multi_tsan_test.cpp"), os.path.join(CSRC, "kz_multi.cpp"), os.path.join(CSRC, "kz_plan.cpp"),
                           "-o", exe, "-pthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok") and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
