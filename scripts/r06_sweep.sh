#!/bin/bash
# round 6: the new large-frame test, then the randomized parity sweeps on the final sources (progress lines keep the call alive)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06k_sweep; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; RC=$?; tail -6 $OUT/tests.log; [ $RC -eq 0 ] || exit $RC
timeout -k 10 900 python scripts/dev/fuzz_sweep.py ${1:-40000} ${2:-3000} --rich > $OUT/fuzz_sweep_rich.log 2>&1 &
P=$!
while kill -0 $P 2>/dev/null; do sleep 45; tail -1 $OUT/fuzz_sweep_rich.log | cut -c1-160; done
wait $P; RC=$?
tail -2 $OUT/fuzz_sweep_rich.log
[ $RC -eq 0 ] || exit $RC
timeout -k 10 600 python scripts/dev/sample_bits.py ${1:-40000} 60 > $OUT/sample_bits.log 2>&1; RC=$?
tail -3 $OUT/sample_bits.log
exit $RC
