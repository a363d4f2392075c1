#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
for w in 8 4; do
for d in 1 9 5 3; do
echo "== waves=$w GL_CONV_HALO_DBG=$d (1 no DMA, 8 no reads, 4 no MFMA, 2 no compute)"; GL_CONV_HALO_WAVES=$w GL_CONV_HALO=1 GL_CONV_HALO_DBG=$d timeout 100 $K tools/halo.shapes 10 conv | grep "^conv 8 64 64 960" | cut -c1-130
done; done
} > gpurun_out/halo4.txt 2>&1
cat gpurun_out/halo4.txt
