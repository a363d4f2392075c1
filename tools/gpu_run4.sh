#!/bin/bash
set -x
mkdir -p gpurun_out
K=gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q ) > gpurun_out/pytest_ops.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_ops.log
tail -15 gpurun_out/pytest_ops.log
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kbench_unet_v3.txt 2>&1
timeout 300 $K tools/vae_b4.shapes 5 - check > gpurun_out/kbench_vae_v3.txt 2>&1
grep -v "^gn\|^ln\|^attn" gpurun_out/kbench_unet_v3.txt | tail -100
tail -12 gpurun_out/kbench_vae_v3.txt
