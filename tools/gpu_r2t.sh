#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call T: halo kernel K-split sweep (GL_CONV_HALO_SPLITS) per shape
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
for sp in 0 1 2 4 8; do
echo "== GL_CONV_HALO_SPLITS=$sp"; GL_CONV_HALO=1 GL_CONV_HALO_SPLITS=$sp timeout 200 $K tools/unet_b8.shapes 10 conv | grep "^conv" | grep "256x" | cut -c1-130
done
} > gpurun_out/halo_splits.txt 2>&1
