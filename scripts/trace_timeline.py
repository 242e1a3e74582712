#!/usr/bin/env python3
"""Timeline facts from rocprofv3 --kernel-trace CSVs (one or more processes on ONE GPU): per kernel name the count, the summed and average
duration; per queue the busy time; and over all processes the union of the kernel intervals (the time at least one kernel was resident),
the sum of durations / union (average number of kernels resident at once) and the gaps in which NO kernel was resident.
    python scripts/trace_timeline.py gpurun_out/r03b/trace1 [more dirs ...] [--from-kernel kz_wf_generate --skip 2]"""
import csv, glob, json, re, sys, collections

def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]

def load(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), d, int(r["Queue_Id"])))
    return rows

def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = sorted(sum((load(d) for d in dirs), []))
    kz = [r for r in rows if r[2].startswith("kz_wf") or r[2].startswith("kz_film")]
    # the timed region: from the first kz kernel to the last one of the wavefront pipeline (drop the stats / megakernel tail of rank 0)
    t0 = kz[0][0]
    mega = [r[0] for r in rows if "megakernel" in r[2] or r[2].startswith("kz_wf_shade<true") or r[2].startswith("kz_wf_count")]
    t1 = min(mega) if mega else kz[-1][1]
    kz = [r for r in kz if r[1] <= t1]
    t1 = max(r[1] for r in kz)
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n, d, q in kz:
        per[n][0] += 1; per[n][1] += e - s
    ev = sorted([(s, 1) for s, e, *_ in kz] + [(e, -1) for s, e, *_ in kz])
    depth = 0; last = ev[0][0]; union = 0; hist = collections.Counter()
    for t, k in ev:
        if depth > 0: union += t - last
        hist[depth] += t - last
        depth += k; last = t
    tot = sum(e - s for s, e, *_ in kz)
    out = {"dirs": dirs, "span_ms": (t1 - t0) / 1e6, "union_ms": union / 1e6, "sum_of_durations_ms": tot / 1e6, "avg_kernels_resident": tot / max(1, union),
           "ms_with_n_kernels_resident": {str(k): round(v / 1e6, 2) for k, v in sorted(hist.items())},
           "kernels": {n: {"calls": c, "total_ms": round(t / 1e6, 2), "avg_ms": round(t / c / 1e6, 3)} for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])}}
    print(json.dumps(out, indent=1))

main()
