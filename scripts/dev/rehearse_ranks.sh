# bench.py under torch.distributed.run at N = 2 and 4 with every rank on the ONE GPU of the box (KZ_BENCH_DEVICE=0: each rank caps its path state):
# the multi-rank path on the final sources - weak-scaling steps with a constant pass per rank, the pipelined host gather, strong_c5 with both dealings.
R=$PWD; O=$R/gpurun_out/${1:-r05e}; mkdir -p $O
for N in 2 4; do
  KZ_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 2 --warmup 2 > $O/bench_n$N.json 2> $O/bench_n$N.err || { tail -n 30 $O/bench_n$N.err; exit 1; }
  python - $O/bench_n$N.json $N <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], d["end_to_end"]["value"], d["config"]["items_per_pass_per_rank"], d["config"]["sampler_table_spp"], json.dumps(d.get("parity")), json.dumps(d.get("strong_c5"))[:1500])
PY
done
