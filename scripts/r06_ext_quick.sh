#!/bin/bash
# round 6: the GPU suite, then the stage times of the two EXT scenes (+ C3 as the lean reference)
R=$GRAFT_REPO_ROOT; NAME=${1:-r06c_ext}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
if [ -z "$KZ_SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests -m gpu -q ${KZ_TESTSEL:+-k "$KZ_TESTSEL"} > $OUT/tests.log 2>&1; RC=$?; tail -12 $OUT/tests.log; [ $RC -eq 0 ] || exit $RC; fi
for s in ext_materials ext_textured ${KZ_SCENES:-c3}; do timeout -k 10 200 python scripts/probe.py stages --scene $s --spp 256 >> $OUT/stages.txt 2>> $OUT/stages.err || exit 1; done
cat $OUT/stages.txt
