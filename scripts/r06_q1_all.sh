#!/bin/bash
# (KZ_Q1_ARGS=--beside: every picture a second time with KzRenderOpts::shadowBeside = 2)
# round 6: the reference's 22 scene files at their own settings (1920x1080, 4096 spp) through the HIP path on the final sources (progress lines keep the call alive)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/q1_full
timeout -k 10 900 python scripts/dev/q1_full.py --all $KZ_Q1_ARGS > gpurun_out/q1_full/all.log 2>&1 &
P=$!
while kill -0 $P 2>/dev/null; do sleep 30; tail -1 gpurun_out/q1_full/all.log | cut -c1-160; done
wait $P; RC=$?; tail -3 gpurun_out/q1_full/all.log; du -sh gpurun_out/q1_full; exit $RC
