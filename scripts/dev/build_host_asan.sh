#!/bin/sh
# The HOST translation units (scene validation + flattening, the BVH builders, tile dealing + the host merge) with AddressSanitizer + UBSan, linked with the normal device
# objects: nano-kazen_amd/csrc/variants/host_asan/libkazen_mi355x.so. GPU sanitizers do not exist on this pool; this covers the code that runs on the CPU.
#   sh scripts/dev/build_host_asan.sh && sh scripts/dev/run_host_asan.sh
set -e
cd "$(dirname "$0")/../../nano-kazen_amd/csrc"
OUT=variants/host_asan; mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
DEV="--offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fno-slp-vectorize"
for u in kz_render kz_film kz_debug kz_multi kz_host kz_bvh kz_arena kz_plan; do rm -f $OUT/$u.o; done
PIDS=""
for u in kz_render kz_film kz_debug; do hipcc -O3 -std=c++17 -fPIC -ffp-contract=off $DEV -c $u.hip -o $OUT/$u.o & PIDS="$PIDS $!"; done
for u in kz_multi kz_host kz_bvh kz_arena kz_plan; do hipcc $FLAGS -fsanitize=address,undefined -fno-omit-frame-pointer -c $u.cpp -o $OUT/$u.o & PIDS="$PIDS $!"; done
for p in $PIDS; do wait $p || { echo "build failed: a translation unit did not compile"; exit 1; }; done
hipcc -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libkazen_mi355x.so $OUT/kz_render.o $OUT/kz_film.o $OUT/kz_debug.o $OUT/kz_multi.o $OUT/kz_host.o $OUT/kz_bvh.o $OUT/kz_arena.o $OUT/kz_plan.o -pthread
rm -f $OUT/*.o
echo "built $(pwd)/$OUT/libkazen_mi355x.so"
