#!/bin/bash
# the randomized parity sweep alone (scripts/r06_sweep.sh without the test suite): bash scripts/r06_sweep_only.sh <first seed> <scenes>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06k_sweep; mkdir -p $OUT; cd $R
timeout -k 10 1100 python scripts/dev/fuzz_sweep.py ${1:-70000} ${2:-6000} --rich > $OUT/fuzz_sweep_rich_$1.log 2>&1 &
P=$!
while kill -0 $P 2>/dev/null; do sleep 45; tail -1 $OUT/fuzz_sweep_rich_$1.log | cut -c1-160; done
wait $P; RC=$?
tail -2 $OUT/fuzz_sweep_rich_$1.log
exit $RC
