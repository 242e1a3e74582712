for t in "" "refill=24" "refill=32" "refill=48" "refill=56" "refill=64" "batch=64" "batch=256" "batch=512" "postpone=16" "postpone=32"; do echo "$t $(timeout -k 10 120 python scripts/probe.py stages --scene c4 --spp 128 --tune "$t" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); st=d['stages_one_pass_alone']; print('%.1f bounce %.2f shadow %.2f'%(d['Msamples_per_s'],st['trace_bounce'],st['trace_shadow']))")"; done
