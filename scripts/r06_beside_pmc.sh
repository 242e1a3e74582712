#!/bin/bash
# counters of the per-lane traversal kernels on a job, one stream (why do the shadow and closest-hit kernels of the q1 asset gain from running beside each other and those of C3 / C4 not?)
#   bash scripts/r06_beside_pmc.sh q1 1920 1080 64
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06v; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
export KZ_SHADOW_BESIDE=1
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$1_$i -- python3 $R/scripts/dev/beside_trace.py "$@" > $OUT/pmc_$1_$i.log 2>&1 || { echo "pmc $C failed"; tail -n 5 $OUT/pmc_$1_$i.log; }
done
python3 - $OUT $1 <<'PY'
import csv, glob, sys, collections, os, json
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
summ = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    d = dict(v)
    d["lanes_per_valu"] = round(d.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, d.get("SQ_ACTIVE_INST_VALU", 1)), 1)
    d["valu_active_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), 4)
    d["valu_busy_simd"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(1, d.get("SQ_BUSY_CYCLES", 1)), 4)
    d["l1_hit"] = round(1 - d.get("TCP_TCC_READ_REQ_sum", 0) / max(1, d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1)), 3)
    summ[k] = d
    print("%-30s %s" % (k, "  ".join("%s=%.4g" % (c.replace("SQ_", "").replace("_sum", ""), x) for c, x in sorted(d.items()))))
json.dump(summ, open(out + "/pmc_%s_summary.json" % tag, "w"), indent=1)
PY
rm -rf $OUT/pmc_$1_*/
