#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call A: pipelined d = 40 attention (attn2_kernel) -- op tests, per-shape A/B against the unpipelined kernel
# (GL_ATTN_V2 = 0 / 1 / 2), VALU issue-cost microbenchmark, short path tests, whole-path bench A/B
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
K=gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error" $O/pytest_ops.log | cut -c1-300
for v in 0 1 2; do
  echo "== GL_ATTN_V2=$v" >> $O/attn_kbench.txt
  GL_ATTN_V2=$v timeout 120 $K tools/attn.shapes 20 >> $O/attn_kbench.txt 2>&1
done
cat $O/attn_kbench.txt | grep -v "^TOTAL\|^shape"
timeout 120 gligen_amd/build/valubench > $O/valubench.txt 2>&1
cat $O/valubench.txt
( timeout 900 python -m pytest tests/test_path_gpu.py tests/test_configs_gpu.py -m gpu -q -x ) > $O/pytest_path.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error" $O/pytest_path.log | cut -c1-300
{
for arm in 0 1 2; do
  echo "== GL_ATTN_V2=$arm"
  GL_ATTN_V2=$arm timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'launches', d.get('launches_per_unet_eval'), 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k) for k in r['kernels'][:6]]"
done
} > $O/attn_bench_ab.txt 2>&1
cat $O/attn_bench_ab.txt
