#!/bin/bash
# round 3: the in-tree build against a variant built from the previous sources (scripts/build_variant.sh prev, before the edit), same call; parity suite on the tree build first
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${KZ_OUT:-r03v}; mkdir -p $OUT
cd $R
V=${KZ_VARIANT:-prev}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_textures.py -m gpu -x -q > $OUT/pytest_tree.log 2>&1; rc=$?; tail -2 $OUT/pytest_tree.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest_tree.log; exit $rc; }
for rep in 1 2; do
for lib in tree $V; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages_$V.txt
  done
done
done
