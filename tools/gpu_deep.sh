#!/bin/bash
mkdir -p gpurun_out
K=gligen_amd/build/kbench
for f in 12,5 12,4 10,5 10,4; do
GL_GEMM_AUTOTUNE=0 KB_FORCE=$f,0 timeout 300 $K tools/unet_b8.shapes 5 - check > gpurun_out/kb_d_${f/,/}.txt 2>&1
echo "== force $f"; grep "MISMATCH\|CHECK\|TOTAL all\|error" gpurun_out/kb_d_${f/,/}.txt | head -9
done
timeout 900 $K tools/unet_b8.shapes 5 - sweep > gpurun_out/sweep_unet_d.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_unet_d.txt | cut -c1-150
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_d_auto.txt 2>&1
grep "TOTAL\|CHECK" gpurun_out/kb_d_auto.txt
