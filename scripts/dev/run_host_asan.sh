#!/bin/sh
# The CPU test suite's host-side cases (scene validation, BVH builds, tile dealing, the host merges, the ABI checks) against the ASan + UBSan build of the host units.
cd "$(dirname "$0")/../.."
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
export KZ_LIB_PATH=$PWD/nano-kazen_amd/csrc/variants/host_asan/libkazen_mi355x.so
export LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_abi_cpu.py tests/test_plan_cpu.py tests/test_xmlscene.py tests/test_shard_gloo.py tests/test_textures.py tests/test_output.py -x -q -m "not gpu" "$@"
