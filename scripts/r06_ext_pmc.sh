#!/bin/bash
# round 6: counters of the shade kernel on the EXT scenes (one PMC group per run; the program directly after `--`)
R=$GRAFT_REPO_ROOT; NAME=${1:-r06e_ext_pmc}; OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in ${KZ_SCENES:-ext_materials ext_textured c3}; do
 for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${s}_$N -- python3 $R/scripts/probe.py stages --scene $s --spp 64 > $OUT/pmc_${s}_$N.log 2>&1 || { echo "pmc $s $C failed"; tail -3 $OUT/pmc_${s}_$N.log; }
 done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/pmc_*/")):
    scene = os.path.basename(d.rstrip("/")).split("_SQ")[0].replace("pmc_", "")
    agg = res.setdefault(scene, collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
import json
summ = {}
for scene, agg in res.items():
    print("==", scene)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]:
        d = dict(v)
        d["lanes_per_valu"] = round(d.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, d.get("SQ_ACTIVE_INST_VALU", 1)), 1)
        d["valu_busy_of_wave_cycles"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), 3)
        summ.setdefault(scene, {})[k] = d
        print("%-42s %s" % (k, "  ".join("%s=%.4g" % (c.replace("SQ_", ""), x) for c, x in sorted(d.items()))))
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_*/
