#!/bin/sh
# Builds a development variant of the library into nano-kazen_amd/csrc/variants/<name>/libkazen_mi355x.so with extra hipcc flags,
# WITHOUT touching the in-tree product build (probes select it with KZ_LIB_PATH). The .so files are git-ignored but travel with gpurun.
#   scripts/build_variant.sh lanestat -DKZ_LANESTAT
set -e
NAME=$1; shift
SRC="$(cd "$(dirname "$0")/../nano-kazen_amd/csrc" && pwd)"
OUT=$SRC/variants/$NAME
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $FLAGS --offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fno-slp-vectorize "$@" -c $SRC/kz_device.hip -o $OUT/kz_device.o
hipcc $FLAGS "$@" -c $SRC/kz_host.cpp -o $OUT/kz_host.o
hipcc $FLAGS "$@" -c $SRC/kz_bvh.cpp -o $OUT/kz_bvh.o
hipcc -shared -fPIC -o $OUT/libkazen_mi355x.so $OUT/kz_device.o $OUT/kz_host.o $OUT/kz_bvh.o -pthread
rm -f $OUT/*.o
echo "built $OUT/libkazen_mi355x.so"
