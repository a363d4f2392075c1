#!/bin/bash
set -x
mkdir -p gpurun_out
K=gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q ) > gpurun_out/pytest_ops.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_ops.log
timeout 300 $K tools/unet_b8.shapes 10 attn > gpurun_out/kbench_attn.txt 2>&1
cat gpurun_out/kbench_attn.txt
tail -5 gpurun_out/pytest_ops.log
