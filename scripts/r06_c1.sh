#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06h_c1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
KZ_C1_MEGA=1 python3 $R/scripts/dev/c1_trace.py > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/dev/c1_trace.py > $OUT/traced.txt 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - $f > $OUT/launches.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: kernels after the last kz_wf_generate
gi = [i for i, r in enumerate(rows) if "kz_wf_generate" in r["Kernel_Name"]]
last = rows[gi[-1]:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
    busy += e - s; prev_end = max(prev_end, e)
print("span %.1f us, kernels %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(last)))
PY
tail -n 3 $OUT/launches.txt
rm -rf $OUT/trace
