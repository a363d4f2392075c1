#!/bin/bash
# one gpurun call: parity tests, kernel micro-bench (old vs LDS-DMA main loop), bench + rocprof stats
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
K=gligen_amd/build/kbench
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
GL_GEMM_VARIANT=0 timeout 300 $K tools/unet_b8.shapes 10 > gpurun_out/kbench_unet_v0.txt 2>&1
GL_GEMM_VARIANT=1 timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kbench_unet_v1.txt 2>&1
GL_GEMM_VARIANT=1 timeout 300 $K tools/vae_b4.shapes 5 - check > gpurun_out/kbench_vae_v1.txt 2>&1
GL_GEMM_VARIANT=0 timeout 300 $K tools/vae_b4.shapes 5 > gpurun_out/kbench_vae_v0.txt 2>&1
tail -8 gpurun_out/kbench_unet_v0.txt gpurun_out/kbench_unet_v1.txt gpurun_out/kbench_vae_v1.txt
tail -5 gpurun_out/pytest_gpu.log
