#!/bin/bash
# round 3: suite after the dead-code removal + the full-size kernel-equivalence test; bench.py --strong (C5) as a 4-rank rehearsal on one GPU
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03m; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -2 $OUT/pytest.log
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=4 KZ_BENCH_DEVICE=0
pids=""
for r in 0 1 2 3; do RANK=$r LOCAL_RANK=$r timeout -k 10 600 python3 bench.py --gpus 4 --strong --no-cpu-baseline > $OUT/s4_$r.json 2> $OUT/s4_$r.err & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "rank failed"; tail -5 $OUT/s4_*.err; exit 1; }; done
grep -h "^rank" $OUT/s4_*.err | cut -c1-400
python3 -c "
import json; d=json.load(open('$OUT/s4_0.json')); print('strong N=4 (one GPU)', d['value'], d['ms_per_step'], d['scaling'], d['end_to_end'], d['config']['workload'][:200])
"
