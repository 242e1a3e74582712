#!/bin/sh
# bench.py (C4, two passes in flight) on the in-tree build and on development variants, alternating, one gpurun call:  sh scripts/dev/inflight_ab.sh <outdir> tree prev ...
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do for lib in "$@"; do
  if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
  timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>$O/bench_$lib.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'pass in flight', r['pass_ms_in_flight'], 'alone', r['pass_ms_alone'], 'first_call_ms', d['config'].get('first_call_ms'))" | tee -a $O/inflight.txt
done; done
