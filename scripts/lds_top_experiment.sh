#!/bin/bash
# Runs on the GPU box: the LDS top-of-tree experiment. Stage times of one C4 pass for a grid of
# (ldsStack, ldsTop) and the L1 (TCP) access counters of the two ends, program directly after `--`.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lds_top
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in "ldsStack=16" "ldsStack=12" "ldsStack=12,ldsTop=21" "ldsStack=12,ldsTop=85" "ldsStack=16,ldsTop=85" "ldsStack=10,ldsTop=341" "ldsStack=8,ldsTop=341" "ldsStack=8"; do
  python3 $R/scripts/probe.py stages --tune $t 2>/dev/null
done > $OUT/stages.jsonl
cat $OUT/stages.jsonl
for t in "ldsStack=12" "ldsStack=12,ldsTop=85"; do
  N=$(echo $t | tr ',=' '__')
  timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_$N -- python3 $R/scripts/probe.py stages --tune $t --spp 64 > $OUT/pmc_$N.log 2>&1 || echo "pmc $t failed"
done
python3 - <<'PY'
import csv, glob, json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/lds_top"
res = {}
for d in sorted(glob.glob(out + "/pmc_*/")):
    acc = {}
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("kz_wf_trace<"):
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res[os.path.basename(d.rstrip("/"))] = acc
json.dump(res, open(out + "/tcp_counters.json", "w"), indent=1)
for k, v in res.items():
    for kk, c in v.items():
        print(k, kk, {n: "%.3g" % x for n, x in c.items()})
PY
