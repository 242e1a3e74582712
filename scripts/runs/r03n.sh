#!/bin/bash
# round 3: BVH4 packets with binary16 planes read through v_fma_mix_f32 (variants/fp16, 96-B nodes, 11-bit planes) against the 64-B byte packets, same call
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03n; mkdir -p $OUT
cd $R
KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/fp16/libkazen_mi355x.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_fp16.log 2>&1; rc=$?; tail -2 $OUT/pytest_fp16.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest_fp16.log; exit $rc; }
for rep in 1 2; do
for lib in tree fp16; do
  for sc in c4 c3; do
    if [ $lib = fp16 ]; then export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/fp16/libkazen_mi355x.so; else unset KZ_LIB_PATH; fi
    echo "$lib $sc $(timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/stages.txt
  done
done
done
for lib in tree fp16; do
  if [ $lib = fp16 ]; then export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/fp16/libkazen_mi355x.so; else unset KZ_LIB_PATH; fi
  echo "$lib $(timeout -k 10 300 python scripts/probe.py counters --scene c4 --spp 32 2>> $OUT/stages.err | tail -1)" | tee -a $OUT/counters.txt
done
