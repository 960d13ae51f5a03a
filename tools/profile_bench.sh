#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of the default bench command,
# then separate PMC passes (never combined with tracing domains other than --kernel-trace).
# Usage: bash tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# headline workload only (both precisions): no child processes under the profiler, no other configurations in the statistics
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_under_trace.log 2>&1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
