#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call P: wide GEGLU kernel whose epilogue touches neither LDS nor global loads behind the next item's first DMA (row
# statistics finalised into registers before it, coefficients carried from the item's own begin) against the previous commit
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3p
mkdir -p $O
B=gligen_amd/build
cp gligen_amd/libgligen_amd.so $B/libgligen_amd.main.so
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "geglu or ln_folded or linear" ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_ops.log | cut -c1-300 | head
{
for round in 1 2; do
for arm in main old; do
  k=$B/kbench; [ $arm != main ] && k=$B/var_$arm/kbench
  echo "== $arm, round $round"
  timeout 200 $k tools/unet_b8.shapes 10 gemm check | grep "^gemm [0-9]* [0-9]* [0-9]* 1 \|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-110
done
done
} > $O/wide_nodrain_kbench.txt 2>&1
cat $O/wide_nodrain_kbench.txt
{
for arm in main old main old; do
  echo "== $arm"
  if [ $arm = old ]; then cp $B/var_old/libgligen_amd.so gligen_amd/libgligen_amd.so; else cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so; fi
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'dominant', r['kernel'], r['kernels'][0]['ms'], round(r['achieved'],1), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so
} > $O/wide_nodrain_bench_ab.txt 2>&1
cat $O/wide_nodrain_bench_ab.txt
