#!/bin/sh
# VGPRs / SGPRs / spills / scratch / LDS / occupancy of every kernel of the product library (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../nano-kazen_amd/csrc"
for u in kz_render kz_film kz_debug; do hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fno-slp-vectorize ${KZ_EXTRA_HIPFLAGS} -Rpass-analysis=kernel-resource-usage -c $u.hip -o /tmp/kz_res_$u.o 2>&1; done |
python3 -c '
import re, sys, subprocess
cur = None; rows = {}
for l in sys.stdin:
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*Function Name: (\S+)", l)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for n, (k, r) in zip(names, rows.items()):
    if "kz_" in n:
        print("%-62s VGPR %3d  SGPR %3d  spill v%-3d s%-3d scratch %4d  LDS %6d  occ %d" % (n.split("(")[0].replace("void ", "")[:62], r.get("VGPRs", -1), r.get("TotalSGPRs", r.get("SGPRs", -1)),
              r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0), r.get("ScratchSize", 0), r.get("LDS Size", 0), r.get("Occupancy", 0)))
'
