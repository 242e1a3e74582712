#!/bin/bash
# round 6: GPU tests (optional selection), then a same-call A/B of build variants on the EXT scenes
R=$GRAFT_REPO_ROOT; NAME=${1:-r06f_ab}; VARIANTS=${2:-tree,nosort}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT; cd $R
if [ -z "$KZ_SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests -m gpu -q ${KZ_TESTSEL:+-k "$KZ_TESTSEL"} > $OUT/tests.log 2>&1; RC=$?; tail -12 $OUT/tests.log; [ $RC -eq 0 ] || exit $RC; fi
timeout -k 10 600 python scripts/probe.py ab --variants $VARIANTS --scenes ${KZ_SCENES:-ext_materials,ext_textured} --reps 2 --spp 256 --out $NAME 2>&1 | tail -30
