#!/bin/bash
# bench.py with every pass of the timed steps on one stream / with its shadow rays beside / as two halves / as the library decides, alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06v; mkdir -p $OUT
for rep in 1 2; do for mode in "--shadow-beside 1 --pass-halves 1" "--shadow-beside 2 --pass-halves 1" "--shadow-beside 1 --pass-halves 2" ""; do
  tag=$(echo "m$mode" | tr -d ' -'); 
  timeout -k 10 400 python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-parity --no-ext-scenes --no-cold-job --no-asset-scene $mode > $OUT/bench_${tag}_$rep.json 2> $OUT/bench_${tag}_$rep.err || { tail -n 5 $OUT/bench_${tag}_$rep.err; exit 1; }
  python3 - $OUT/bench_${tag}_$rep.json "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-40s value %8.1f  ms/step %7.2f  %s" % (sys.argv[2] or "(defaults)", d["value"], d["ms_per_step"], d["config"].get("how_the_last_timed_pass_ran")), flush=True)
PY
done; done
