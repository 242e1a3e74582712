#!/bin/bash
# round 3: shared beam lists (one per replica for the whole pixel set) - suite, full renders, then the end-state profile
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03q; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -2 $OUT/pytest.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
rm -rf gpurun_out/prof_${KZ_PROF_TAG:-r03q}
bash scripts/runs/r03l.sh
