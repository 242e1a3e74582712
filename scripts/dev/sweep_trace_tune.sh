# development helper: whole-call throughput and stage times of C4 (and C3) under KzTuning variants, three repetitions each, one gpurun call
for rep in 1 2 3; do for t in ${KZ_TUNES:-"" "batch=192" "batch=256" "batch=384" "batch=512"}; do for sc in c4 c3; do echo "$sc [$t] $(timeout -k 10 120 python scripts/probe.py stages --scene $sc --spp 256 --tune "$t" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stages_one_pass_alone']; print('%.1f bounce %.2f shadow %.2f'%(d['Msamples_per_s'],st['trace_bounce'],st['trace_shadow']))")"; done; done; done
