#!/bin/bash
# v5 unified GEMM (buffer-load DMA): check vs variant 1 in auto mode and with each tile forced, then sweep
mkdir -p gpurun_out
K=gligen_amd/build/kbench
GL_GEMM_VARIANT=4 timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_u_auto.txt 2>&1
tail -8 gpurun_out/kb_u_auto.txt
for f in 8,5 8,4 4,5 4,4 2,5 2,4; do
GL_GEMM_VARIANT=4 KB_FORCE=$f,0 timeout 300 $K tools/unet_b8.shapes 5 - check > gpurun_out/kb_u_${f/,/}.txt 2>&1
echo "== force $f"; grep "MISMATCH\|CHECK\|TOTAL all\|error" gpurun_out/kb_u_${f/,/}.txt | head -8
done
GL_GEMM_VARIANT=4 timeout 300 $K tools/vae_b4.shapes 3 - check > gpurun_out/kb_u_vae.txt 2>&1
tail -4 gpurun_out/kb_u_vae.txt
GL_GEMM_VARIANT=4 timeout 900 $K tools/unet_b8.shapes 5 - sweep > gpurun_out/sweep_unet_u.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_unet_u.txt | cut -c1-160 | head -100
GL_GEMM_VARIANT=4 timeout 400 $K tools/vae_b4.shapes 3 - sweep > gpurun_out/sweep_vae_u.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_vae_u.txt | cut -c1-160
