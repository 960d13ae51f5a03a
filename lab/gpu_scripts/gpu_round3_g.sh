#!/bin/bash
# round 3, rocprofv3 passes of the last build: headline bench command (kernel trace + PMC traffic / cycle counters), FastDVDnet and
# FFDNet + DDnet fp32 iterations, the fp32 finetune event.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r03f
bash tools/profile_bench.sh $T > gpurun_out/${T}_profile_bench.log 2>&1; tail -30 gpurun_out/${T}_profile_bench.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
prof() {  # tag, env assignments..., then script
  local tag=$1; shift
  local out=$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_$tag
  mkdir -p $out
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/$SCRIPT > $out/trace.log 2>&1 )
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $out > $out/summary.txt 2>&1
  echo "== $tag"; tail -1 $out/trace.log; grep -v "at::native\|rocclr" $out/summary.txt | cut -c1-200 | head -12
}
SCRIPT=fastdvd_bench.py prof fastdvd_f32 SCIPNP_CONV_PRECISION=f32 SCIPNP_STREAMS=1
SCRIPT=ddnet_bench.py prof ddnet_f32 SCIPNP_CONV_PRECISION=f32 SCIPNP_STREAMS=1
SCRIPT=finetune_bench.py prof ffdnet_finetune_f32 SCIPNP_CONV_PRECISION=f32 FT_REPS=3
