#!/bin/bash
# round 5: kernel-trace statistics of the fp32 FastDVDnet and FFDNet + DDnet iterations (one stream) -- the PixelShuffle-store layers
set -u
T=${1:-r05zx}
cd /tmp && export TMPDIR=/tmp SCIPNP_STREAMS=1 SCIPNP_CONV_PRECISION=f32 FD_STEPS=4 DD_STEPS=4
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_fd $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_dd
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_fd -- python3 $GRAFT_REPO_ROOT/tools/fastdvd_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${T}_fastdvd_f32.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_dd -- python3 $GRAFT_REPO_ROOT/tools/ddnet_bench.py > $GRAFT_REPO_ROOT/gpurun_out/${T}_ddnet_f32.log 2>&1
cd $GRAFT_REPO_ROOT
grep "ms/iteration" gpurun_out/${T}_fastdvd_f32.log gpurun_out/${T}_ddnet_f32.log
python3 tools/trace_stats.py gpurun_out/prof_${T}_fd 6 | cut -c1-130
python3 tools/trace_stats.py gpurun_out/prof_${T}_dd 6 | cut -c1-130
find gpurun_out/prof_${T}_fd gpurun_out/prof_${T}_dd -name "*.csv" -size +1M -delete
