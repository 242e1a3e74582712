#!/bin/bash
R=$GRAFT_REPO_ROOT; NAME=${1:-r06d_shadestat}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
for s in ext_materials ext_textured c3; do echo "== $s" >> $OUT/shadestat.txt; KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/shadestat/libkazen_mi355x.so timeout -k 10 200 python scripts/probe.py shadestat --scene $s --spp 64 >> $OUT/shadestat.txt 2>&1 || exit 1; done
cat $OUT/shadestat.txt
