#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call I: conv gather with per-pixel tap masks (no per-tile coordinate rebuild) + in-kernel split-K fold (agent-scope
# stores, ticket per tile): op tests, kbench conv / gemm A/B of the fold on one box, in-situ A/B, suite, bench
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
K=gligen_amd/build/kbench
for f in 0 1; do
  echo "== GL_GEMM_SPLIT_FOLD=$f"
  GL_GEMM_SPLIT_FOLD=$f timeout 300 $K tools/unet_b8.shapes 10 - check | grep "^TOTAL\|CHECK\|MISMATCH" | cut -c1-160
  GL_GEMM_SPLIT_FOLD=$f timeout 300 python tools/insitu.py 2>/dev/null | head -1
done > gpurun_out/fold_ab.txt 2>&1
cat gpurun_out/fold_ab.txt
timeout 300 $K tools/unet_b8.shapes 10 conv | grep "^conv" | cut -c1-120 > gpurun_out/conv_i.txt
head -30 gpurun_out/conv_i.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
for f in 0 1; do
GL_GEMM_SPLIT_FOLD=$f timeout 400 python bench.py --steps 3 --no-cpu-baseline > gpurun_out/bench_i$f.json 2> gpurun_out/bench_i$f.err
cut -c1-200 gpurun_out/bench_i$f.json; tail -2 gpurun_out/bench_i$f.err | cut -c1-300
done
