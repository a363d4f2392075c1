#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call B: GPU tests (all), fused-QKV A/B (in-situ per-problem profile with the fusion off / on), C4 bench
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
GL_QKV_FUSED=0 timeout 300 python tools/insitu.py > gpurun_out/insitu_qkv0.txt 2> gpurun_out/insitu_qkv0.err
head -1 gpurun_out/insitu_qkv0.txt
timeout 300 python tools/insitu.py > gpurun_out/insitu_qkv1.txt 2> gpurun_out/insitu_qkv1.err
head -1 gpurun_out/insitu_qkv1.txt; grep "true>" gpurun_out/insitu_qkv1.txt
timeout 400 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/bench_qkv1.json 2> gpurun_out/bench_qkv1.err
cut -c1-200 gpurun_out/bench_qkv1.json; tail -2 gpurun_out/bench_qkv1.err | cut -c1-300
timeout 400 python bench.py --config C4 --steps 2 --no-cpu-baseline > gpurun_out/bench_C4.json 2> gpurun_out/bench_C4.err
cut -c1-330 gpurun_out/bench_C4.json; tail -2 gpurun_out/bench_C4.err | cut -c1-300
