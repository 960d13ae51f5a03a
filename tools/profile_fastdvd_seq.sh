#!/bin/bash
# ON THE GPU BOX: in-order kernel trace of one FastDVDnet iteration (single stream), 512x512x8
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fdseq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SCIPNP_STREAMS=${SCIPNP_STREAMS:-1} FD_STEPS=2 SCIPNP_CONV_PRECISION=${SCIPNP_CONV_PRECISION:-f16x3}
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/fastdvd_bench.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_sequence.py $OUT/trace 45 > $OUT/sequence.txt 2>&1
cat $OUT/sequence.txt
