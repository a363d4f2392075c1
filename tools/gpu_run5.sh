#!/bin/bash
set -x
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kbench_unet_v3b.txt 2>&1
grep -v "^gn\|^ln\|^attn" gpurun_out/kbench_unet_v3b.txt | awk '{print $NF, $0}' | cut -c1-120 | tail -75 | head -66 > /dev/null
tail -9 gpurun_out/kbench_unet_v3b.txt
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
cat gpurun_out/bench2.json
