#!/bin/bash
# bench.py with the shadow kernels of every pass in front of (--shadow-beside 1) / beside (2) the closest-hit kernel, alternating
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06v; mkdir -p $OUT
for rep in 1 2; do for mode in 1 2; do
  timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-ext-scenes --no-cold-job --shadow-beside $mode > $OUT/bench_sb${mode}_$rep.json 2> $OUT/bench_sb${mode}_$rep.err || { tail -n 5 $OUT/bench_sb${mode}_$rep.err; exit 1; }
  python3 - $OUT/bench_sb${mode}_$rep.json $mode <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d.get("reference_scene") or {}
print("--shadow-beside %s  value %8.1f  ms/step %7.2f   reference_scene %s / beside %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("value"), (r.get("shadow_beside") or {}).get("value")), flush=True)
PY
done; done
