#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call Q: halo kernel v3 (activation fragments of the next tile read ahead, reads interleaved with the MFMA rows): 4 vs 8 waves
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
echo "== default"; timeout 200 $K tools/unet_b8.shapes 10 conv | grep "^conv\|^TOTAL conv" | cut -c1-130
for w in 4 8; do
echo "== GL_CONV_HALO=1 waves=$w check"; GL_CONV_HALO_WAVES=$w GL_CONV_HALO=1 timeout 300 $K tools/unet_b8.shapes 10 conv check | grep "^conv\|^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-160
done
echo "== vae default"; timeout 200 $K tools/vae_b4.shapes 5 conv | grep "^conv\|^TOTAL conv" | cut -c1-130
for w in 4 8; do
echo "== vae GL_CONV_HALO=1 waves=$w check"; GL_CONV_HALO_WAVES=$w GL_CONV_HALO=1 timeout 300 $K tools/vae_b4.shapes 5 conv check | grep "^conv\|^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-160
done
for w in 4 8; do
for d in 0 1 2 5 3; do
echo "== waves=$w GL_CONV_HALO_DBG=$d (1 no DMA, 2 no compute, 4 no MFMA)"; GL_CONV_HALO_WAVES=$w GL_CONV_HALO=1 GL_CONV_HALO_DBG=$d timeout 100 $K tools/halo.shapes 10 conv | grep "^conv" | cut -c1-130
done; done
} > gpurun_out/halo3.txt 2>&1
grep "==\|TOTAL\|CHECK\|MISM" gpurun_out/halo3.txt | head -24
