#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# kbench (autotuned) + GPU tests + bench
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_at.txt 2>&1
grep "^gn\|^TOTAL\|CHECK\|MISMATCH" gpurun_out/kb_at.txt | cut -c1-120
timeout 300 $K tools/vae_b4.shapes 5 - check > gpurun_out/kb_at_vae.txt 2>&1
tail -6 gpurun_out/kb_at_vae.txt
( timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
tail -3 gpurun_out/bench.err
