#!/bin/bash
# v5 wide GEMM: forced 256x160 / 256x128 on every shape they fit (bit-compare against variant 1), then the tile/split sweep
mkdir -p gpurun_out
K=gligen_amd/build/kbench
GL_GEMM_VARIANT=4 KB_FORCE=8,5,0 timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_w85.txt 2>&1
tail -8 gpurun_out/kb_w85.txt
GL_GEMM_VARIANT=4 KB_FORCE=8,4,0 timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_w84.txt 2>&1
tail -8 gpurun_out/kb_w84.txt
GL_GEMM_VARIANT=4 timeout 900 $K tools/unet_b8.shapes 5 - sweep > gpurun_out/sweep_unet_w.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_unet_w.txt | cut -c1-160 | head -100
GL_GEMM_VARIANT=4 timeout 400 $K tools/vae_b4.shapes 3 - sweep > gpurun_out/sweep_vae_w.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_vae_w.txt | cut -c1-160
