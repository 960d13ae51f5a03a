#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel trace of one online-finetune iteration (tools/finetune_bench.py); FT_DENOISER=fastdvd for FastDVDnet
set -u
TAG=${1:-ft}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/finetune_bench.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "at::native\|rocclr" $OUT/summary.txt | cut -c1-220 | head -16
