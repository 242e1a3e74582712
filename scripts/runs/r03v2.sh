#!/bin/bash
# round 3: the in-tree build against several variants (KZ_VARIANTS), whole-call throughput only (two passes in flight), three repetitions; parity subset on the tree build first
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${KZ_OUT:-r03v2}; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_textures.py -m gpu -x -q > $OUT/pytest_tree.log 2>&1; rc=$?; tail -1 $OUT/pytest_tree.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest_tree.log; exit $rc; }
for rep in 1 2 3; do
for lib in tree ${KZ_VARIANTS:-prev}; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages.txt
  done
done
done
