#!/bin/bash
# round 3, second GPU call: stream-priority policies of the in-process passes; the 4-process rehearsal of round 2 (one GPU shared by 4 ranks),
# plain and under rocprofv3 --kernel-trace, next to the 1-process bench under the same trace (timeline comparison)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03b; mkdir -p $OUT
cd $R
for m in 1 2; do
  timeout -k 10 300 python scripts/probe.py sched --scene c4 --spp 1024 --tune streamPriority=$m --values '2,0,27;3,0,27;4,0,27;8,0,27' > $OUT/sched_prio$m.jsonl 2> $OUT/sched_prio$m.err || { tail -5 $OUT/sched_prio$m.err; exit 1; }
  echo "streamPriority=$m"; cat $OUT/sched_prio$m.jsonl
done
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 WORLD_SIZE=4 KZ_BENCH_DEVICE=0
pids=""
for r in 0 1 2 3; do RANK=$r LOCAL_RANK=$r timeout -k 10 400 python3 bench.py --gpus 4 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/b4_$r.json 2> $OUT/b4_$r.err & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "rank failed"; tail -3 $OUT/b4_*.err; exit 1; }; done
cat $OUT/b4_0.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
export MASTER_PORT=29512
pids=""
for r in 0 1 2 3; do RANK=$r LOCAL_RANK=$r timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace4_$r -- python3 $R/bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/t4_$r.json 2> $OUT/t4_$r.err & pids="$pids $!"; done
for p in $pids; do wait $p || { echo "traced rank failed"; tail -3 $OUT/t4_*.err; exit 1; }; done
cut -c1-300 $OUT/t4_0.json
unset MASTER_ADDR MASTER_PORT WORLD_SIZE KZ_BENCH_DEVICE
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/t1.json 2> $OUT/t1.err || { tail -3 $OUT/t1.err; exit 1; }
cut -c1-300 $OUT/t1.json
du -sh $OUT
