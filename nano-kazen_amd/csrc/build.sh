#!/bin/sh
# Builds libkazen_mi355x.so for gfx950 (cross-compiles without a GPU). In-tree so it travels with gpurun.
set -e
# -fno-slp-vectorize (device code): left to itself the compiler packs the cross products of the triangle test and of the shading code into
# v_pk_mul_f32 / v_pk_add_f32 with register shuffles around them - a packed f32 instruction issues no faster than its two halves on gfx950
# (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever") and the shuffles are pure overhead: C4 1 588 -> 1 662 Msamples/s, same results
# (profiles/r03s_no_slp). The one place where packing pays, the BVH4 node step, uses explicit two-lane vector FMAs (kz_devfn.h node4Keys).
cd "$(dirname "$0")"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
hipcc $FLAGS --offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fno-slp-vectorize ${KZ_EXTRA_HIPFLAGS} -c kz_device.hip -o kz_device.o
hipcc $FLAGS -c kz_host.cpp -o kz_host.o
hipcc $FLAGS -c kz_bvh.cpp -o kz_bvh.o
hipcc -shared -fPIC -o libkazen_mi355x.so kz_device.o kz_host.o kz_bvh.o -pthread
echo "built $(pwd)/libkazen_mi355x.so"
