#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call N: conv halo kernel -- correctness (kbench check against the v1 kernel) and per-shape timing against the default
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
{
echo "== default"; timeout 200 $K tools/unet_b8.shapes 10 conv | grep "^conv\|^TOTAL conv" | cut -c1-130
for h in 1; do
echo "== GL_CONV_HALO=$h check"; GL_CONV_HALO=$h timeout 300 $K tools/unet_b8.shapes 10 conv check | grep "^conv\|^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-160
done
echo "== vae default"; timeout 200 $K tools/vae_b4.shapes 5 conv | grep "^conv\|^TOTAL conv" | cut -c1-130
echo "== vae GL_CONV_HALO=1 check"; GL_CONV_HALO=1 timeout 300 $K tools/vae_b4.shapes 5 conv check | grep "^conv\|^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-160
} > gpurun_out/halo1.txt 2>&1
grep "==\|TOTAL\|CHECK\|MISM" gpurun_out/halo1.txt
