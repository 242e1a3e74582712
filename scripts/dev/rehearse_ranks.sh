R=$PWD; O=$R/gpurun_out/r04f; mkdir -p $O
python -c "import __graft_entry__ as g" 2>/dev/null
timeout -k 10 300 python scripts/probe.py dealing --scene c5 --spp 256 --reps 2 "--values=-1,0;0,0" > $O/dealing_c5.txt 2>&1; tail -n 3 $O/dealing_c5.txt | cut -c1-300
for N in 2 4; do
  KZ_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 2 --warmup 1 > $O/bench_n$N.json 2> $O/bench_n$N.err || { tail -n 20 $O/bench_n$N.err; exit 1; }
  python -c "import json,sys; d=json.load(open('$O/bench_n$N.json')); print($N, d['value'], d['end_to_end']['value'], json.dumps(d.get('strong_c5'))[:900])"
done
