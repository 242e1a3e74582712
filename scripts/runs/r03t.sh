#!/bin/bash
# round 3: compiler variants against the in-tree build (now -fno-slp-vectorize): slp (the old default), -O2, max-ilp / max-memory-clause scheduling; same call
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03t; mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in tree slp o2 maxilp maxmem; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages.txt
  done
done
done
