#!/bin/bash
# round 5: band kernel v2 + one-round-trip dual-update/projection kernel -- parity tests touching the TV step, then the ADMM-TV
# iteration's kernel durations (rocprofv3 kernel trace) for: new kernels, old band kernel, old dual-project kernel
set -u
# (the SCIPNP_TV_BAND_V1 / SCIPNP_DUAL_PROJECT_GENERAL rows need the laboratory build: make -C adaptivepnp_sci_amd/csrc tvvariant NAME=lab TVFLAGS=-DSCIPNP_LAB_SWITCHES)
export SCIPNP_LIB=${SCIPNP_LIB:-$GRAFT_REPO_ROOT/build/variants/libscipnp_tvlab.so}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r05z}
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_gpu_units.py tests/test_gpu_configs.py tests/test_gpu_cabi_host.py -x -q -m gpu -k "tv or TV or admm or units or config" > gpurun_out/${TAG}_tv_tests.txt 2>&1
rc=$?; tail -n 5 gpurun_out/${TAG}_tv_tests.txt
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for v in new oldband oldproj; do
  export SCIPNP_TV_BAND_V1=0 SCIPNP_DUAL_PROJECT_GENERAL=0
  [ $v = oldband ] && export SCIPNP_TV_BAND_V1=1
  [ $v = oldproj ] && export SCIPNP_DUAL_PROJECT_GENERAL=1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_iter_trace.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v.log 2>&1 || exit 1
  echo "== $v"
  python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v 2>&1 | grep -v "at::native\|rocclr" | cut -c1-200 | head -12 | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_${v}_summary.txt
  find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_iter_$v -name "*.csv" -size +1M -delete
done
unset SCIPNP_TV_BAND_V1 SCIPNP_DUAL_PROJECT_GENERAL
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${TAG}_tvsweep
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep_kernels.txt
find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_tvsweep -name "*.csv" -size +1M -delete
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/tv_bench.py 2>&1 | grep "ADMM-TV\|whole" | tee gpurun_out/${TAG}_tv_bench.txt
