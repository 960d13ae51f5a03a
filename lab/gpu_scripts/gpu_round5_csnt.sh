#!/bin/bash
# round 5: the split-fp16 kernel's c8s stores with the nt hint (make csvariant) against the product, f16x3 network iterations
set -u
cd $GRAFT_REPO_ROOT
export SCIPNP_CONV_PRECISION=f16x3
for v in product csnt; do
  unset SCIPNP_LIB
  [ $v != product ] && export SCIPNP_LIB=$GRAFT_REPO_ROOT/build/variants/libscipnp_$v.so
  echo "== $v"
  DD_STEPS=6 timeout -k 10 200 python tools/ddnet_bench.py 2>&1 | grep "ms/iteration"
  FD_STEPS=6 timeout -k 10 200 python tools/fastdvd_bench.py 2>&1 | grep "ms/iteration"
  timeout -k 10 300 python tools/probes/w4nt_crossover.py 2>&1 | grep "256x256x16\|512x512x8\|1024x1024"
done
