#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call O: halo kernel ablation (what bounds a tile: DMA, fragment reads, MFMAs)
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
for d in 0 1 2 4 5 3; do
echo "== GL_CONV_HALO_DBG=$d (1 no DMA, 2 no compute, 4 no MFMA)"; GL_CONV_HALO=1 GL_CONV_HALO_DBG=$d timeout 100 $K tools/halo.shapes 10 conv | grep "^conv" | cut -c1-130
done
} > gpurun_out/halo_abl.txt 2>&1
cat gpurun_out/halo_abl.txt
