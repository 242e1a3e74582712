#!/bin/bash
# round 3, third GPU call: pixel beams for the camera rays (kz_wf_beam + kz_wf_trace_list) against the packet kernel, C4 and C3; the VALU issue
# rate with long loop bodies
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03c; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 scripts/micro/valu_clock 2000 2.0 > $OUT/valu_clock_body16.json 2> $OUT/valu_clock.err || { tail -3 $OUT/valu_clock.err; exit 1; }
timeout -k 10 200 scripts/micro/valu_clock_body2 16000 2.0 > $OUT/valu_clock_body2.json 2>> $OUT/valu_clock.err || { tail -3 $OUT/valu_clock.err; exit 1; }
python3 -c "
import json
for f in ('$OUT/valu_clock_body16.json', '$OUT/valu_clock_body2.json'):
    for r in json.load(open(f))['results']: print(f[-12:], r['op'], r['waves_per_simd'], 'clk', r['shader_clock_GHz_in_kernel'], 'G/s', r['chip_G_wave_instr_per_s_wall'], 'span/med', r['all_waves_span_over_median_wave'])
"
for sc in c4 c3; do
  for t in "packetPrimary=2" "packetPrimary=0" "packetPrimary=0,sppPerPass=256" "packetPrimary=2,sppPerPass=256"; do
    timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 --tune $t >> $OUT/stages_$sc.jsonl 2>> $OUT/stages.err || { tail -5 $OUT/stages.err; exit 1; }
  done
  cat $OUT/stages_$sc.jsonl
done
