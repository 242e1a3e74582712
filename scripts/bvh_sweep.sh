#!/bin/bash
# Same-call sweep of the builder's leaf parameters (KZ_BVH_MAX_LEAF <= 4, KZ_BVH_NODE_COST) on the C4 stage times: scripts/bvh_sweep.sh "4 3 2" "0.3 0.5 0.7"
for ml in ${1:-4}; do for nc in ${2:-0.7}; do
  echo "maxLeaf=$ml nodeCost=$nc $(KZ_BVH_MAX_LEAF=$ml KZ_BVH_NODE_COST=$nc python scripts/probe.py stages 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['Msamples_per_s'], d['stages_one_pass_alone'])")"
done; done
