#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call N: split-K reduce with four slab loads in flight and its bias / per-sample bias / residual / gate requested in front
# of the slabs; LayerNorm kernel by 64-lane chunk count -- against the previous commit's library; then the full GPU suite
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3n
mkdir -p $O
B=gligen_amd/build
cp gligen_amd/libgligen_amd.so $B/libgligen_amd.main.so
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_ops.log | cut -c1-300 | head -20
{
for round in 1 2; do
  echo "== main, round $round"
  timeout 300 $B/kbench tools/unet_b8.shapes 10 - check > $O/kb_main_$round.txt 2>&1; grep "^TOTAL\|CHECK\|MISMATCH\|mismatch" $O/kb_main_$round.txt | cut -c1-150
  echo "== old, round $round"
  timeout 300 $B/var_old/kbench tools/unet_b8.shapes 10 - check > $O/kb_old_$round.txt 2>&1; grep "^TOTAL\|CHECK\|MISMATCH\|mismatch" $O/kb_old_$round.txt | cut -c1-150
done
echo "== per shape (split-K problems and LayerNorms), round 2: main | old (us, rate)"
paste <(grep "^gemm\|^conv\|^ln" $O/kb_main_2.txt | cut -c1-100) <(grep "^gemm\|^conv\|^ln" $O/kb_old_2.txt | cut -c58-75) | grep "/[2-9]\|/1[0-9]\|/2[0-9]\|/3[0-9]\|^ln"
} > $O/reduce_kbench_ab.txt 2>&1
cat $O/reduce_kbench_ab.txt
{
for arm in main old main old; do
  echo "== $arm"
  if [ $arm = old ]; then cp $B/var_old/libgligen_amd.so gligen_amd/libgligen_amd.so; else cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so; fi
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so
} > $O/reduce_bench_ab.txt 2>&1
cat $O/reduce_bench_ab.txt
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "^FAILED|^ERROR|passed|failed|pytest rc" $O/pytest_gpu.log | cut -c1-250
