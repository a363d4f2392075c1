#!/bin/bash
mkdir -p gpurun_out
K=gligen_amd/build/kbench
GL_GEMM_VARIANT=4 timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_u2_auto.txt 2>&1
cat gpurun_out/kb_u2_auto.txt | cut -c1-120
for f in 8,5 4,4 2,5 2,4; do
GL_GEMM_VARIANT=4 KB_FORCE=$f,0 timeout 300 $K tools/unet_b8.shapes 5 - check > gpurun_out/kb_u2_${f/,/}.txt 2>&1
echo "== force $f"; grep "MISMATCH\|CHECK\|TOTAL all\|error" gpurun_out/kb_u2_${f/,/}.txt | head -8
done
GL_GEMM_VARIANT=4 timeout 300 $K tools/vae_b4.shapes 3 - check > gpurun_out/kb_u2_vae.txt 2>&1
tail -4 gpurun_out/kb_u2_vae.txt
