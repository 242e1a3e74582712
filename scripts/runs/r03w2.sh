#!/bin/bash
# round 3, end: bench.py as the driver launches it at N = 2 (torch.distributed.run, both ranks on the one GPU of the box: KZ_BENCH_DEVICE=0) - a rehearsal of the multi-rank path on the final sources
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03w2; mkdir -p $OUT
cd $R
KZ_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${KZ_N:-2} --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus ${KZ_N:-2} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/n2.json 2> $OUT/n2.err || { tail -20 $OUT/n2.err; exit 1; }
grep -h "^rank" $OUT/n2.err | cut -c1-300
python3 -c "
import json; d=json.loads(open('$OUT/n2.json').read().strip().splitlines()[-1]); print('N ranks on one GPU:', d['value'], d['ms_per_step'], d['scaling'], d['end_to_end'])"
