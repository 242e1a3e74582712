#!/bin/bash
# round 3: pixel beams with a capped LDS stack - list statistics, stage times and parity on C4 / C3; VALU issue rate by operand pattern
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03d; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -3 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 true
python3 -c "
import json
for r in json.load(open('$OUT/valu_clock.json'))['results']: print(r['op'], '| waves', r['waves_per_simd'], 'clk', r['shader_clock_GHz_in_kernel'], 'G/s', r['chip_G_wave_instr_per_s_wall'])
"
for sc in c4 c3; do
  timeout -k 10 300 python scripts/probe.py counters --scene $sc --spp 64 --values 'packetPrimary=0;packetPrimary=2' > $OUT/counters_$sc.jsonl 2>> $OUT/stages.err || { tail -5 $OUT/stages.err; exit 1; }
  cat $OUT/counters_$sc.jsonl
  for t in "packetPrimary=2" "packetPrimary=0" "packetPrimary=0,sppPerPass=256" "packetPrimary=2,sppPerPass=256"; do
    timeout -k 10 300 python scripts/probe.py stages --scene $sc --spp 256 --tune $t >> $OUT/stages_$sc.jsonl 2>> $OUT/stages.err || { tail -5 $OUT/stages.err; exit 1; }
  done
  cat $OUT/stages_$sc.jsonl
done
