#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call K: (1) epilogues that wait for their coefficient / residual loads ONCE in front of the first store (head-layout
# epilogues of the q,k,v^T projection, residual prefetch of the staged row-major epilogue) against the previous commit's library
# (var_old); (2) gemm_geglu_kernel with the piece VALU laid between the MFMAs (GL_GEGLU_PIPE=1 vs 0)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3k
mkdir -p $O
B=gligen_amd/build
cp gligen_amd/libgligen_amd.so $B/libgligen_amd.main.so
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_ops.log | cut -c1-300 | head -20
{
for round in 1 2; do
  echo "== main GL_GEGLU_PIPE=0, round $round"
  GL_GEGLU_PIPE=0 timeout 300 $B/kbench tools/unet_b8.shapes 10 gemm check > $O/kb_main0_$round.txt 2>&1; grep "^TOTAL\|CHECK\|MISMATCH\|mismatch" $O/kb_main0_$round.txt | cut -c1-150
  echo "== main GL_GEGLU_PIPE=1, round $round"
  GL_GEGLU_PIPE=1 timeout 300 $B/kbench tools/unet_b8.shapes 10 gemm check > $O/kb_main1_$round.txt 2>&1; grep "^TOTAL\|CHECK\|MISMATCH\|mismatch" $O/kb_main1_$round.txt | cut -c1-150
  echo "== old, round $round"
  timeout 300 $B/var_old/kbench tools/unet_b8.shapes 10 gemm check > $O/kb_old_$round.txt 2>&1; grep "^TOTAL\|CHECK\|MISMATCH\|mismatch" $O/kb_old_$round.txt | cut -c1-150
done
echo "== per shape, round 2: main pipe=0 | main pipe=1 | old (us, TF/s)"
paste <(grep "^gemm" $O/kb_main0_2.txt | cut -c1-75) <(grep "^gemm" $O/kb_main1_2.txt | cut -c58-75) <(grep "^gemm" $O/kb_old_2.txt | cut -c58-75)
} > $O/epi_wait_kbench_ab.txt 2>&1
cat $O/epi_wait_kbench_ab.txt
{
for arm in main1 main0 old main1 main0 old; do
  echo "== $arm"
  if [ $arm = old ]; then cp $B/var_old/libgligen_amd.so gligen_amd/libgligen_amd.so; else cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so; fi
  p=1; [ $arm = main0 ] && p=0
  GL_GEGLU_PIPE=$p timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'dominant', r['kernel'], round(r['achieved'],1), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so
} > $O/epi_wait_bench_ab.txt 2>&1
cat $O/epi_wait_bench_ab.txt
