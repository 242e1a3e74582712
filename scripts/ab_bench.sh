#!/bin/sh
# Same-call A/B of two builds of the library on the bench workload (box-to-box variance is 3-6 %, so only numbers from ONE gpurun call compare):
#   scripts/ab_bench.sh <variant-name> [rounds]     -> alternates variants/<name>/libkazen_mi355x.so and the in-tree build
# Output: gpurun_out/ab_<name>.txt with one "<tag> Msamples/s ms_per_step" line per run.
NAME=$1; ROUNDS=${2:-2}
OUT=gpurun_out/ab_$NAME.txt
mkdir -p gpurun_out; : > $OUT
one() {   # tag, lib path ('' = in-tree)
    KZ_LIB_PATH=$2 python bench.py --steps 3 --warmup 1 2>gpurun_out/ab_$NAME.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" >> $OUT || exit 1
}
i=0
while [ $i -lt $ROUNDS ]; do
    one "$NAME" nano-kazen_amd/csrc/variants/$NAME/libkazen_mi355x.so || exit 1
    one tree "" || exit 1
    i=$((i + 1))
done
cat $OUT
