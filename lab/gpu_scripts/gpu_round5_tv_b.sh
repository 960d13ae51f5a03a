#!/bin/bash
# round 5: band kernel schedule v2 -- parity tests that touch the TV step, then A/B timing of the band kernel (rocprofv3 trace)
set -u
# (the SCIPNP_TV_BAND_V1 / SCIPNP_DUAL_PROJECT_GENERAL rows need the laboratory build: make -C adaptivepnp_sci_amd/csrc tvvariant NAME=lab TVFLAGS=-DSCIPNP_LAB_SWITCHES)
export SCIPNP_LIB=${SCIPNP_LIB:-$GRAFT_REPO_ROOT/build/variants/libscipnp_tvlab.so}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_gpu_units.py tests/test_gpu_configs.py tests/test_gpu_cabi_host.py -x -q -m gpu -k "tv or TV or admm or units or config" > gpurun_out/r05z_tv_tests.txt 2>&1
rc=$?; tail -n 5 gpurun_out/r05z_tv_tests.txt
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v$v
  SCIPNP_TV_BAND_V1=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v$v -- python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v$v.log 2>&1 || exit 1
  echo "== SCIPNP_TV_BAND_V1=$v"
  python3 $GRAFT_REPO_ROOT/tools/probes/tv_band_sweep.py $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v$v | tee $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v${v}_kernels.txt
  find $GRAFT_REPO_ROOT/gpurun_out/r05z_tvsweep_v$v -name "*.csv" -size +1M -delete
done
