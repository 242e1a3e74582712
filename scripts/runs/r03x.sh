#!/bin/bash
# round 3: slot order of the BVH4 children (the order any-hit rays take them) - builder variants (kz_bvh.cpp -DKZ_SLOT_ORDER=k linked with the in-tree device objects) against the in-tree build, same call
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03x; mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in tree ${KZ_VARIANTS:-ord1 ord2 ord3 ord4 ord5 ord6}; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages.txt
  done
done
done
