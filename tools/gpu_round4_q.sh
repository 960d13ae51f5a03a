#!/bin/bash
# round 4: wave priority in the boundary phases of conv3x3_c8w4_kernel (W4_PRIO bit 1: epilogue, bit 2: prologue / first column pass)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" build/variants/libscipnp_prio1.so build/variants/libscipnp_prio2.so build/variants/libscipnp_prio3.so ""; do
  echo "== SCIPNP_LIB=$v"
  SCIPNP_LIB=${v:+$GRAFT_REPO_ROOT/$v} timeout -k 10 200 python tools/wino_bench.py 2>&1 | grep "F(4x4)" | cut -c1-200
done | tee gpurun_out/r04q_prio_ab.txt
