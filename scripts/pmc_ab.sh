#!/bin/bash
# Same-call counter A/B of a variant build vs the in-tree build: one PMC pass each over one bench step (program directly after `--`).
# (--no-cold-job: under --pmc the profiler's preload has initialised the GPU before bench.py starts - bench.py must not start child processes then; it also refuses by itself)
#   scripts/pmc_ab.sh <variant-name> "SQ_INSTS_VALU SQ_INSTS_SALU ..."   -> gpurun_out/pmcab_<name>/{variant,tree}/..., summary printed per kernel
NAME=$1; C=${2:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcab_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$NAME/libkazen_mi355x.so timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/variant -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-asset-scene --no-cold-job --no-ext-scenes --no-parity > $OUT/variant.log 2>&1 || { tail -5 $OUT/variant.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/tree -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-asset-scene --no-cold-job --no-ext-scenes --no-parity > $OUT/tree.log 2>&1 || { tail -5 $OUT/tree.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for side in ("variant", "tree"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(out + "/" + side + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", side)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
        print("%-50s %s" % (k, "  ".join("%s=%.4g" % (c.replace("SQ_", ""), x) for c, x in sorted(v.items()))))
PY
