#!/bin/bash
# Runs on the GPU box (via gpurun): the VALU issue-ceiling microbenchmark, plain and under the same rocprofv3 PMC counters the
# traversal profiles use (each counter group in its own run; the program directly after `--`). Output: gpurun_out/valu_peak/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/valu_peak
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$R/scripts/micro/valu_peak
N2=$(python3 -c "import json;print(json.load(open('$R/scripts/micro/valu_peak_counts.json'))['2']['valu'])")
N3=$(python3 -c "import json;print(json.load(open('$R/scripts/micro/valu_peak_counts.json'))['3']['valu'])")
$B -1 20000 $N2 $N3 > $OUT/plain.json || exit 1
cat $OUT/plain.json
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
    --output-format csv -d $OUT/pmc_sq -- $B -1 20000 $N2 $N3 > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $B -1 20000 $N2 $N3 > $OUT/pmc_grbm.log 2>&1 || echo "pmc grbm failed"
python3 $R/scripts/micro/valu_peak_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
