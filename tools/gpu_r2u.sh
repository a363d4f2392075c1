#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
echo "== default kernel"; GL_CONV_HALO=0 timeout 100 $K tools/halo.shapes 10 conv | grep "^conv" | cut -c1-130
echo "== halo"; GL_CONV_HALO=1 timeout 100 $K tools/halo.shapes 10 conv check | grep "^conv\|CHECK\|MISM" | cut -c1-130
echo "== halo"; GL_CONV_HALO=1 timeout 100 $K tools/halo.shapes 10 conv | grep "^conv" | cut -c1-130
} > gpurun_out/halo5.txt 2>&1
cat gpurun_out/halo5.txt
