#!/bin/bash
mkdir -p gpurun_out
K=gligen_amd/build/kbench
GL_GEMM_VARIANT=2 timeout 600 $K tools/unet_b8.shapes 5 - sweep > gpurun_out/sweep_unet.txt 2>&1
GL_GEMM_VARIANT=2 timeout 300 $K tools/vae_b4.shapes 3 - sweep > gpurun_out/sweep_vae.txt 2>&1
grep "^SWEEP" gpurun_out/sweep_unet.txt gpurun_out/sweep_vae.txt | cut -c1-200
