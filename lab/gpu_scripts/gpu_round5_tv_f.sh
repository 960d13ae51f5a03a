#!/bin/bash
# round 5: ADMM-TV iteration, the product (hand-off state stored non-temporally) against plain stores (make tvnt TVNT=0) and
# non-temporal loads as well (TVNT=2)
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r05zd}
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-product}; do
  unset SCIPNP_LIB
  [ $v != product ] && export SCIPNP_LIB=$GRAFT_REPO_ROOT/build/variants/libscipnp_tv$v.so
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_iter_trace.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v.log 2>&1 || exit 1
  echo "== $v"
  python3 $GRAFT_REPO_ROOT/tools/trace_stats.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v 4 | cut -c1-150 | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_${v}_summary.txt
  find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -name "*.csv" -size +1M -delete
  cd $GRAFT_REPO_ROOT
  timeout -k 10 300 python tools/tv_bench.py 2>&1 | grep "ADMM-TV one-stage 256x256x8 defer=1\|ADMM-TV two-stage 256x256x8 defer=1" | tee gpurun_out/${TAG}_tv_bench_$v.txt
  cd /tmp
done
