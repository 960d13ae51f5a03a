#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04h}
# the variants come from `make -C adaptivepnp_sci_amd/csrc variants` (run it before gpurun: the .so files travel with the snapshot)
for f in build/variants/libscipnp_vblk4.so build/variants/libscipnp_vblk12.so; do [ -f "$f" ] || { echo "missing $f: run make -C adaptivepnp_sci_amd/csrc variants" >&2; exit 1; }; done
for v in "" build/variants/libscipnp_vblk4.so build/variants/libscipnp_vblk12.so ""; do
  echo "== SCIPNP_LIB=$v"
  SCIPNP_LIB=${v:+$GRAFT_REPO_ROOT/$v} timeout -k 10 200 python tools/wino_bench.py 2>&1 | grep "F(4x4)" | cut -c1-200
done | tee gpurun_out/${T}_vblk_ab.txt
echo done
