#!/bin/bash
mkdir -p gpurun_out
K=gligen_amd/build/kbench
for f in 4,5 8,5 2,5 4,4 8,4 2,4; do
GL_GEMM_AUTOTUNE=0 KB_FORCE=$f,0 timeout 300 $K tools/unet_b8.shapes 5 - check > gpurun_out/kb_u3_${f/,/}.txt 2>&1
echo "== force $f"; grep "MISMATCH\|CHECK\|TOTAL\|error" gpurun_out/kb_u3_${f/,/}.txt | head -9
done
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_u3_auto.txt 2>&1
grep "^gemm\|^conv\|TOTAL\|CHECK" gpurun_out/kb_u3_auto.txt | cut -c1-120
timeout 300 $K tools/vae_b4.shapes 3 - check > gpurun_out/kb_u3_vae.txt 2>&1
tail -4 gpurun_out/kb_u3_vae.txt
