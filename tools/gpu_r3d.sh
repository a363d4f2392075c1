#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call D: LayerNorms folded into their consumer GEMMs (row statistics from the producer's epilogue): parity suite + A/B
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3d
mkdir -p $O
( timeout 900 python -m pytest tests/test_path_gpu.py tests/test_configs_gpu.py -m gpu -q -x ) > $O/pytest_path.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|assert" $O/pytest_path.log | cut -c1-400 | head -20
{
for arm in 0 1 0 1; do
  echo "== GL_LN_FOLD=$arm"
  GL_LN_FOLD=$arm timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'launches', d.get('launches_per_unet_eval'), 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k) for k in r['kernels'][:14]]"
done
} > $O/lnfold_bench_ab.txt 2>&1
cat $O/lnfold_bench_ab.txt
