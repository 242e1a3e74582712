#!/bin/bash
# round 3: per-kernel times of the camera stage with pixel beams (kernel trace of one probe run on C4 and C3)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sc in c4 c3; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$sc -- python3 $R/scripts/probe.py stages --scene $sc --spp 128 --opts passes_in_flight=1 > $OUT/stages_$sc.json 2> $OUT/stages_$sc.err || { tail -5 $OUT/stages_$sc.err; exit 1; }
  cat $OUT/stages_$sc.json
  python3 - $OUT/trace_$sc <<'PY'
import csv, glob, sys, re
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        if n.startswith("kz_"): print("%-44s calls %4s avg %9.3f ms total %9.2f ms" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
