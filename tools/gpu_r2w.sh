#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call W: wide GEMM kernel (256-row tiles, 8 waves, one workgroup per CU) against gemm_u_kernel per shape
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
echo "== default"; timeout 200 $K tools/unet_b8.shapes 10 gemm | grep "^gemm\|^TOTAL gemm" | cut -c1-130
for w in 4 10; do
echo "== GL_GEMM_WIDE=$w check"; GL_GEMM_WIDE=$w timeout 300 $K tools/unet_b8.shapes 10 gemm check | grep "^gemm\|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-160
done
} > gpurun_out/wide1.txt 2>&1
grep "==\|TOTAL\|CHECK\|MISM" gpurun_out/wide1.txt
