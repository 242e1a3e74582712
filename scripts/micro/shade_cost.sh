#!/bin/sh
# Static VALU cost of the pieces of a shaded hit: compiles shade_cost.hip to gfx950 ISA and prints, per piece, the instruction counts minus the empty kernel's.
cd "$(dirname "$0")"
hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fgpu-flush-denormals-to-zero -S --cuda-device-only shade_cost.hip -o /tmp/shade_cost.s 2>/dev/null || exit 1
python3 - <<'PY'
import re
s = open('/tmp/shade_cost.s').read()
# SIMD cycles per wave64 instruction at 4-8 waves/SIMD, measured by valu_peak.hip / op_cost.hip (profiles/valu_peak.json, profiles/op_cost.json)
SLOW = r'v_(?:mul_lo|mul_hi|mad_u64|mad_i64|cvt_|div_scale|div_fmas|div_fixup|ldexp|frexp|min|max|med3|cmp|cndmask|and_or|perm|bfe|bfi|\w+_f64|mul_u32_u24|mad_u32_u24|readlane|readfirstlane|writelane)'
TRANS = r'v_(?:rcp|rsq|sqrt|exp|log|sin|cos)_'
def cost(b):
    c = 0.0
    for l in b.splitlines():
        t = l.strip().split()
        if not t: continue
        op = t[0]
        if op.startswith('v_'):
            c += 8.6 if re.match(TRANS, op) else (4.5 if re.match(SLOW, op) else 2.9)
        elif op.startswith('s_setreg'): c += 3.5
    return c
rows = {}
for m in re.finditer(r'\n(piece_\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S):
    b = m.group(2)
    cnt = lambda pat: len(re.findall(pat, b, re.M))
    rows[m.group(1)] = dict(valu=cnt(r'^\s+v_\w+'), salu=cnt(r'^\s+s_(?!waitcnt|nop|endpgm|branch|cbranch)\w+'), div=cnt(r'^\s+v_div_fixup_f32'), trans=cnt(r'^\s+' + TRANS),
                            mul32=cnt(r'^\s+v_mul_(?:lo|hi)_u32|^\s+v_mad_u64_u32'), vmem=cnt(r'^\s+(?:global|buffer|flat)_load'), branch=cnt(r'^\s+s_cbranch'), cyc=cost(b))
e = rows['piece_empty']
print('%-26s %6s %6s %5s %6s %6s %5s %7s %9s' % ('piece', 'VALU', 'SALU', 'div', 'trans', 'mul32', 'vmem', 'branch', 'VALU cyc'))
for k, r in rows.items():
    print('%-26s %6d %6d %5d %6d %6d %5d %7d %9.0f' % (k[6:], r['valu'] - e['valu'], r['salu'] - e['salu'], r['div'], r['trans'], r['mul32'], r['vmem'] - e['vmem'], r['branch'], r['cyc'] - e['cyc']))
PY
