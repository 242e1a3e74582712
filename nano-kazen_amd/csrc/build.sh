#!/bin/sh
# Builds libkazen_mi355x.so for gfx950 (cross-compiles without a GPU). In-tree so it travels with gpurun.
set -e
cd "$(dirname "$0")"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $FLAGS --offload-arch=gfx950 -fgpu-flush-denormals-to-zero ${KZ_EXTRA_HIPFLAGS} -c kz_device.hip -o kz_device.o
hipcc $FLAGS -c kz_host.cpp -o kz_host.o
hipcc $FLAGS -c kz_bvh.cpp -o kz_bvh.o
hipcc -shared -fPIC -o libkazen_mi355x.so kz_device.o kz_host.o kz_bvh.o -pthread
echo "built $(pwd)/libkazen_mi355x.so"
