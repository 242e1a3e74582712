#!/bin/bash
# round 3: -fno-slp-vectorize (the compiler packs the cross products of the triangle test and of the shading code into v_pk_mul / v_pk_add with register shuffles) against the
# default build, same call; parity suite on the variant first
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03s; mkdir -p $OUT
cd $R
V=${KZ_VARIANT:-noslp}
KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$V/libkazen_mi355x.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_textures.py -m gpu -x -q > $OUT/pytest_$V.log 2>&1; rc=$?; tail -2 $OUT/pytest_$V.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest_$V.log; exit $rc; }
for rep in 1 2; do
for lib in tree $V; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages_$V.txt
  done
done
done
