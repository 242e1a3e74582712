#!/bin/bash
# round 3: list kernel with wave-uniform (scalar) list walks against the previous build, same call; parity suites first
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03p; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -2 $OUT/pytest.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
for rep in 1 2; do
for lib in tree prev; do
  for sc in c4 c3; do
    if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages.txt
  done
done
done
