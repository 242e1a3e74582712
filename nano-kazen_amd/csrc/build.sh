#!/bin/sh
# Builds libkazen_mi355x.so for gfx950 (cross-compiles without a GPU). In-tree so it travels with gpurun.
set -e
# -fno-slp-vectorize (device code): left to itself the compiler packs the cross products of the triangle test and of the shading code into
# v_pk_mul_f32 / v_pk_add_f32 with register shuffles around them - a packed f32 instruction issues no faster than its two halves on gfx950
# (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever") and the shuffles are pure overhead: C4 1 588 -> 1 662 Msamples/s, same results
# (profiles/r03s_no_slp). The one place where packing pays, the BVH4 node step, uses explicit two-lane vector FMAs (kz_devfn.h node4Keys).
# usage: build.sh [output directory]   (default: this directory; scripts/build_variant.sh passes variants/<name>); extra compiler flags in KZ_EXTRA_HIPFLAGS
cd "$(dirname "$0")"
OUT=${1:-.}
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
DEV="--offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fno-slp-vectorize ${KZ_EXTRA_HIPFLAGS}"
# the translation units compile side by side (kz_state.h says what lives where); every job's exit status is checked and no object of an earlier
# build can stand in for one that failed to compile now
DEVICE_UNITS="kz_render kz_film kz_debug"
HOST_UNITS="kz_multi kz_host kz_bvh kz_arena kz_plan"
for u in $DEVICE_UNITS $HOST_UNITS; do rm -f "$OUT/$u.o"; done
PIDS=""
for u in $DEVICE_UNITS; do hipcc $FLAGS $DEV -c $u.hip -o "$OUT/$u.o" & PIDS="$PIDS $!"; done
for u in $HOST_UNITS; do hipcc $FLAGS ${KZ_EXTRA_HIPFLAGS} -c $u.cpp -o "$OUT/$u.o" & PIDS="$PIDS $!"; done
FAILED=0
for p in $PIDS; do wait $p || FAILED=1; done
[ $FAILED -eq 0 ] || { echo "build failed: a translation unit did not compile"; exit 1; }
OBJS=""
for u in $DEVICE_UNITS $HOST_UNITS; do [ -f "$OUT/$u.o" ] || { echo "build failed: $u"; exit 1; }; OBJS="$OBJS $OUT/$u.o"; done
hipcc -shared -fPIC -o "$OUT/libkazen_mi355x.so" $OBJS -pthread
rm -f $OBJS
echo "built $(cd "$OUT" && pwd)/libkazen_mi355x.so"
