#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call S: halo conv kernel as the default for M >= 2048 -- op tests, whole-path A/B (GL_CONV_HALO=0 vs default), full suite
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
{
for arm in 0 8 0 8; do
  echo "== GL_CONV_HALO=$arm"
  GL_CONV_HALO=$arm timeout 300 python tools/insitu.py 2>/dev/null | head -1
  GL_CONV_HALO=$arm timeout 400 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'vae ms', round(d['vae_decode_ms'],2), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
} > gpurun_out/halo_bench_ab.txt 2>&1
cat gpurun_out/halo_bench_ab.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-250
