#!/bin/bash
# round 3 freeze: full renders of the BASELINE configs, then bench + kernel stats + PMC passes (scripts/profile_bench.sh)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03l; mkdir -p $OUT
cd $R
timeout -k 10 600 python scripts/probe.py configs c5 > $OUT/configs.log 2> $OUT/configs.err || { tail -5 $OUT/configs.err; exit 1; }
cp gpurun_out/configs.json $OUT/configs_full_renders.json; cat $OUT/configs.log | cut -c1-400
bash scripts/profile_bench.sh ${KZ_PROF_TAG:-r03l}
