#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call C: kbench (bit/tolerance check against the v2 kernel) + GPU tests + in-situ profile + bench, after the in-kernel split-K reduce
export TMPDIR=/tmp
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_r2c.txt 2>&1
grep "^TOTAL\|CHECK\|MISMATCH" gpurun_out/kb_r2c.txt | cut -c1-160
( timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
timeout 300 python tools/insitu.py > gpurun_out/insitu_r2c.txt 2> gpurun_out/insitu_r2c.err
head -1 gpurun_out/insitu_r2c.txt; grep "split-K" gpurun_out/insitu_r2c.txt | cut -c1-150
timeout 400 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err
cut -c1-200 gpurun_out/bench_r2c.json; tail -2 gpurun_out/bench_r2c.err | cut -c1-300
