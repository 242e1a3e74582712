#!/bin/bash
# kernel traces of one job with the shadow rays of a bounce in front of (KzRenderOpts::shadowBeside = 1) / beside (2) its closest-hit rays: bash scripts/r06_beside_trace.sh q1 1920 1080 64
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06v; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for mode in 1 2; do
  export KZ_SHADOW_BESIDE=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$mode -- python3 $R/scripts/dev/beside_trace.py "$@" > $OUT/traced_$1_$mode.txt 2>&1 || { tail -n 20 $OUT/traced_$1_$mode.txt; exit 1; }
  f=$(find $OUT/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 - $f > $OUT/launches_$1_$2x$3x$4_$mode.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gi = [i for i, r in enumerate(rows) if "kz_wf_generate" in r["Kernel_Name"]]
last = rows[gi[-1]:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %7.1f  dur %9.1f  end %9.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
    busy += e - s; prev_end = max(prev_end, e)
print("span %.1f us, kernels %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(last)))
PY
  tail -n 1 $OUT/launches_$1_$2x$3x$4_$mode.txt; cat $OUT/traced_$1_$mode.txt
  rm -rf $OUT/trace_$mode
done
