#!/bin/bash
# round 6: where the EXT scenes spend their time - per-stage device times (probe.py stages) and a kernel trace of the same two scenes
R=$GRAFT_REPO_ROOT; NAME=${1:-r06b_ext}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
for s in ext_materials ext_textured c3; do timeout -k 10 200 python scripts/probe.py stages --scene $s --spp 256 >> $OUT/stages.txt 2>> $OUT/stages.err || exit 1; done
cat $OUT/stages.txt
cd /tmp && export TMPDIR=/tmp
for s in ext_materials ext_textured; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$s -- python3 $R/scripts/probe.py stages --scene $s --spp 256 > $OUT/trace_$s.log 2>&1 || { tail -5 $OUT/trace_$s.log; exit 1; }
  f=$(find $OUT/trace_$s -name "*kernel_stats.csv" | head -1); echo "== $s"; head -12 $f | cut -c1-200
  cp $f $OUT/kernel_stats_$s.csv; rm -rf $OUT/trace_$s
done
