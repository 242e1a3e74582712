#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes (each counter group in its own run, as the
# MI355X guide prescribes; the program directly after `--`). Outputs under gpurun_out/prof_$1/. The profiled runs say how a pass runs (one stream: what C4 keeps) instead of
# letting the replica time its first large passes four ways - the kernel statistics then hold one kind of pass.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$KZ_STATS_ONLY" ] || python3 $R/bench.py --steps 6 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asset-scene --no-cold-job --no-ext-scenes --no-parity --profile-pass --shadow-beside 1 --pass-halves 1 > $OUT/stats.log 2>&1 || exit 1
[ -n "$KZ_STATS_ONLY" ] && exit 0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-asset-scene --no-cold-job --no-ext-scenes --no-parity --profile-pass --shadow-beside 1 --pass-halves 1 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
  echo "pmc $C done"
done
