#!/bin/bash
# round 6: the cold one-frame jobs alone, three rounds (upload_s / render_s / download_s per job)
R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/${1:-r06t_cold}; mkdir -p $OUT
for i in 1 2 3; do for w in c4 q1; do sleep 4; python bench.py --cold-job $w 2>/dev/null | tail -1 >> $OUT/cold.txt; done; done
python3 - $OUT/cold.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d["workload"][:12], d["value"], "upload", d["upload_s"], "render", d["render_s"], "download", d["download_s"], "total", d["seconds"])
PY
