#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call I: single-read GroupNorm (gn_fused_kernel): op tests incl. the replay test, per-shape A/B, whole path A/B with two
# batches in flight (two contexts' kernels waiting at the same time), parity suite
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3i
mkdir -p $O
K=gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm" ) > $O/pytest_gn.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_gn.log | cut -c1-300 | head -20
{
for f in 0 1 0 1; do
  echo "== GL_GN_FUSED=$f"
  GL_GN_FUSED=$f timeout 200 $K tools/unet_b8.shapes 10 gn | grep "^gn\|^TOTAL gn" | cut -c1-110
done
} > $O/gn_fused_kbench.txt 2>&1
grep "==\|TOTAL\|gn 8 4096 320 0 1\|gn 8 1024 640 0 1" $O/gn_fused_kbench.txt
{
for f in 0 1 0 1; do
  echo "== GL_GN_FUSED=$f"
  GL_GN_FUSED=$f timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>$O/bench_$f.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'launches', d.get('launches_per_unet_eval'), 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k) for k in r['kernels'] if 'gn_' in k['kernel']]"
done
} > $O/gn_fused_bench_ab.txt 2>&1
cat $O/gn_fused_bench_ab.txt; tail -2 $O/bench_1.err | cut -c1-300
( timeout 1200 python -m pytest tests/test_configs_gpu.py tests/test_parity_gpu.py -m gpu -q -x ) > $O/pytest_parity.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|^E  " $O/pytest_parity.log | cut -c1-300 | head
