#!/bin/sh
# Builds a development variant of the library into nano-kazen_amd/csrc/variants/<name>/libkazen_mi355x.so with extra hipcc flags,
# WITHOUT touching the in-tree product build (probes select it with KZ_LIB_PATH). The .so files are git-ignored but travel with gpurun.
#   scripts/build_variant.sh lanestat -DKZ_LANESTAT
set -e
NAME=$1; shift
SRC="$(cd "$(dirname "$0")/../nano-kazen_amd/csrc" && pwd)"
KZ_EXTRA_HIPFLAGS="$*" sh "$SRC/build.sh" "$SRC/variants/$NAME"
