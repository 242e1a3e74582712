#!/bin/sh
# kz_wf_beam (and the camera-ray kernels) of the in-tree build and of development variants, from rocprofv3 --kernel-trace --stats of one short C4 / C3 render each
#   sh scripts/dev/beam_time.sh <outdir under gpurun_out> tree prev ...
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do for sc in c4 c3; do
  if [ $lib = tree ]; then unset KZ_LIB_PATH; else export KZ_LIB_PATH=$R/nano-kazen_amd/csrc/variants/$lib/libkazen_mi355x.so; fi
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${lib}_$sc -- python3 $R/scripts/probe.py stages --scene $sc --spp 64 --opts passes_in_flight=1 > $O/${lib}_$sc.log 2>&1
  f=$(find $O/${lib}_$sc -name "*kernel_stats.csv" | head -n 1)
  python3 - "$f" $lib $sc <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('void ', '')
    if 'beam' in n or 'trace_list' in n or 'trace_packet' in n or 'tiles_expand' in n:
        print(sys.argv[2], sys.argv[3], "%-34s calls %3s avg %8.3f ms" % (n[:34], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done; done
