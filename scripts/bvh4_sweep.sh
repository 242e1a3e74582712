#!/bin/bash
# Same-call comparison of the BVH4 collapse (KZ_BVH4_COLLAPSE=0 greedy / 1 SAH-optimal DP) and its leaf parameters on the C4 stage times and
# executed counters: scripts/bvh4_sweep.sh "0.4 0.6 0.9" "4 8"
run() { echo "$1 $(env $1 python scripts/probe.py stages 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['Msamples_per_s'], d['stages_one_pass_alone'])")"; }
run "KZ_BVH4_COLLAPSE=0"
for pc in ${1:-0.6}; do for ml in ${2:-4}; do run "KZ_BVH4_COLLAPSE=1 KZ_BVH4_PRIM_COST=$pc KZ_BVH4_MAX_LEAF=$ml"; done; done
run "KZ_BVH4_COLLAPSE=0"
