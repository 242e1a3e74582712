#!/bin/sh
# Regenerates tests/golden/int_kats.json from the reference's own hash.h / pcg32.h (oracle/kat_ref_main.cpp) and from
# the random:: block of its common.cpp (oracle/kat_ref_permute.cpp) and from the Fresnel functions of the same file (oracle/kat_ref_fresnel.cpp: fp32 bit patterns) and from struct DiscretePDF of its dpdf.h + the power-of-4 helpers of its common.h (oracle/kat_ref_dpdf.cpp). Only works where /root/reference exists (this
# container); the JSON it writes is the committed fixture that travels to the GPU box.
set -e
cd "$(dirname "$0")/../../oracle"
make ref
T=$(mktemp -d); trap 'rm -rf "$T"' EXIT
./_ref/kat_ref > $T/a.json
./_ref/kat_ref_permute > $T/b.json
./_ref/kat_ref_fresnel > $T/c.json
./_ref/kat_ref_dpdf > $T/d.json
python3 -c "
import json
a = json.load(open('$T/a.json')); a.update(json.load(open('$T/b.json'))); a.update(json.load(open('$T/c.json'))); a.update(json.load(open('$T/d.json')))
open('../tests/golden/int_kats.json', 'w').write(json.dumps(a, indent=None, separators=(',', ':')).replace('],[', '],\n[').replace('},{', '},\n{') + '\n')
"
echo "wrote tests/golden/int_kats.json"
