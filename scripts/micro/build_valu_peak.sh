#!/bin/bash
# Builds scripts/micro/valu_peak for gfx950 (cross-compiles without a GPU) and counts, from the disassembly, the VALU
# instructions in the timed loop of every variant -> valu_peak_counts.json (read by valu_peak.sh on the GPU box).
set -e
cd "$(dirname "$0")"
T=$(mktemp -d)
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off valu_peak.hip -o valu_peak --save-temps=obj -o $T/valu_peak
cp $T/valu_peak valu_peak
python3 - "$T" <<'EOF'
import glob, json, re, sys
from collections import Counter
s = open(glob.glob(sys.argv[1] + "/*gfx950*.s")[0]).read()
out = {}
for k in range(5):  # the five loop variants (the single-opcode kernels are asm, counted by construction)
    m = re.search(r"^_Z9valu_loopILi%dE\w*:[^\n]*\n(.*?)^\.Lfunc_end" % k, s, re.S | re.M)
    parts = re.split(r"^(\.LBB\d+_\d+):.*$", m.group(1), flags=re.M)
    for i in range(1, len(parts) - 1, 2):
        lab, txt = parts[i], parts[i + 1]
        if re.search(r"s_cbranch_\w+ %s\b" % re.escape(lab), txt):
            ins = [l.split()[0] for l in (x.strip() for x in txt.splitlines()) if l and not l.startswith((";", "."))]
            v = [x for x in ins if x.startswith("v_")]
            out[str(k)] = {"valu": len(v), "all": len(ins), "mix": dict(Counter(v).most_common())}
json.dump(out, open("valu_peak_counts.json", "w"), indent=1)
print(json.dumps({k: (v["valu"], v["all"]) for k, v in out.items()}))
EOF
rm -rf $T
