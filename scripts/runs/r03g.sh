#!/bin/bash
# round 3: full GPU suite with the tile gather / dynamic dealing / ABI v4, bench.py N=1, a 4-rank rehearsal on one GPU (end_to_end), the
# experiments variant
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03g; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?; grep -E "C5 gather|passed|failed" $OUT/pytest.log | tail -5
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
timeout -k 10 600 python3 bench.py --steps 6 --warmup 1 > $OUT/bench1.json 2> $OUT/bench1.err || { tail -5 $OUT/bench1.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/bench1.json')); r=d['roofline']
print('N=1', d['value'], d['ms_per_step'], d['end_to_end'], r['stages_ms_one_pass_alone'], r['pass_ms_alone'], r['pass_ms_in_flight'], r.get('counter_facts_withheld'))
print(d['cpu_baseline'])
print(r['hbm_model']['counters_per_sample_executed'])
"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 WORLD_SIZE=4 KZ_BENCH_DEVICE=0
pids=""
for r in 0 1 2 3; do RANK=$r LOCAL_RANK=$r timeout -k 10 400 python3 bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b4_$r.json 2> $OUT/b4_$r.err & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "rank failed"; tail -3 $OUT/b4_*.err; exit 1; }; done
grep -h "^rank" $OUT/b4_*.err
python3 -c "
import json; d=json.load(open('$OUT/b4_0.json')); print('N=4 (one GPU)', d['value'], d['ms_per_step'], d['end_to_end'])
"
