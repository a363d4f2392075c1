#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call X: wide GEMM kernel for the GEGLU projections -- op tests, whole-path A/B (GL_GEMM_WIDE=0 vs default), full suite
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
{
for arm in 0 1 0 1; do
  echo "== GL_GEMM_WIDE=$arm"
  GL_GEMM_WIDE=$arm timeout 300 python tools/insitu.py 2>/dev/null | head -1
  GL_GEMM_WIDE=$arm timeout 400 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'vae ms', round(d['vae_decode_ms'],2), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
} > gpurun_out/wide_bench_ab.txt 2>&1
cat gpurun_out/wide_bench_ab.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
