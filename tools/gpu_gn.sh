#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
export TMPDIR=/tmp
mkdir -p gpurun_out/gnprof
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/gnprof
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $K $S 5 "gn 8" ) > $OUT.log 2>&1
find $OUT -name "*kernel_trace.csv" | head -2
