#!/bin/bash
# round 5: ADMM-TV iteration -- clock stamps of the band kernel, kernel durations of the iteration (new / old band / old projection)
set -u
# (the SCIPNP_TV_BAND_V1 / SCIPNP_DUAL_PROJECT_GENERAL rows need the laboratory build: make -C adaptivepnp_sci_amd/csrc tvvariant NAME=lab TVFLAGS=-DSCIPNP_LAB_SWITCHES)
export SCIPNP_LIB=${SCIPNP_LIB:-$GRAFT_REPO_ROOT/build/variants/libscipnp_tvlab.so}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r05z}
SCIPNP_LIB=$GRAFT_REPO_ROOT/build/variants/libscipnp_tvstamps.so timeout -k 10 200 python tools/probes/tv_band_stamps.py 2>&1 | tee gpurun_out/${TAG}_band_stamps.txt
cd /tmp && export TMPDIR=/tmp
for v in new oldband oldproj; do
  export SCIPNP_TV_BAND_V1=0 SCIPNP_DUAL_PROJECT_GENERAL=0
  [ $v = oldband ] && export SCIPNP_TV_BAND_V1=1
  [ $v = oldproj ] && export SCIPNP_DUAL_PROJECT_GENERAL=1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_iter_trace.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v.log 2>&1 || exit 1
  echo "== $v"
  python3 $GRAFT_REPO_ROOT/tools/trace_stats.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v 8 | cut -c1-170 | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_${v}_summary.txt
  find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -name "*.csv" -size +1M -delete
done
unset SCIPNP_TV_BAND_V1 SCIPNP_DUAL_PROJECT_GENERAL
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep_kernels.txt
find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep -name "*.csv" -size +1M -delete
