#!/bin/bash
# round 3, first GPU call: parity suite on the refactored launch code, the clock / issue-rate microbenchmark, the pass-schedule matrix on C4
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03a; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 scripts/micro/valu_clock 20000 2.0 > $OUT/valu_clock.json 2> $OUT/valu_clock.err || { tail -3 $OUT/valu_clock.err; exit 1; }
cat $OUT/valu_clock.json
timeout -k 10 600 python scripts/probe.py sched --scene c4 --spp 1024 --values '2,0,27;1,0,27;3,0,27;4,0,27;6,0,27;8,0,27;2,256,27;4,256,27;2,1024,27;4,1024,27;8,1024,27;4,0,26;8,0,26;8,0,25;4,1024,26;8,1024,26;2,0,27' > $OUT/sched.jsonl 2> $OUT/sched.err || { tail -5 $OUT/sched.err; exit 1; }
cat $OUT/sched.jsonl
