#!/bin/bash
# round 6: GPU suite, then a bench run; logs under gpurun_out/$1
NAME=${1:-r06a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q ${KZ_TESTSEL:+-k "$KZ_TESTSEL"} > $OUT/tests.log 2>&1; RC=$?
tail -15 $OUT/tests.log
[ $RC -eq 0 ] || exit $RC
shift
timeout -k 10 700 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.log; RC=$?
tail -5 $OUT/bench.log
python3 -c "
import json,sys
d=json.load(open('$OUT/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','parity')})
print('cold', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('cold_job') or {}).items()})
print('ext', {k:(v.get('value'), v.get('cpu_oracle',{}).get('value')) for k,v in (d.get('ext_scenes') or {}).items() if isinstance(v,dict)})
print('ref', (d.get('reference_scene') or {}).get('value'))
"
exit $RC
