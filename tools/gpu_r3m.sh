#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call M: unconditional (clamped-address) loads in the GroupNorm / LayerNorm kernels, the head-layout and staged-residual
# epilogue changes of call L in their final form (no bias2 in the row-GEMM instantiations) against the previous commit's library
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3m
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
echo "== per shape, round 2: main | old (us, rate)"
paste <(grep "^gemm\|^gn\|^ln" $O/kb_main_2.txt | cut -c1-75) <(grep "^gemm\|^gn\|^ln" $O/kb_old_2.txt | cut -c58-75)
echo "== VAE decoder shapes: main, old"
timeout 300 $B/kbench tools/vae_b4.shapes 5 - check | tail -8
timeout 300 $B/var_old/kbench tools/vae_b4.shapes 5 - check | tail -8
} > $O/loads_kbench_ab.txt 2>&1
cat $O/loads_kbench_ab.txt
{
for arm in main old main old main old; do
  echo "== $arm"
  if [ $arm = old ]; then cp $B/var_old/libgligen_amd.so gligen_amd/libgligen_amd.so; else cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so; fi
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k['kernel'], k['ms']) for k in r['kernels'] if 'gn_' in k['kernel'] or 'ln_' in k['kernel']]"
done
cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so
} > $O/loads_bench_ab.txt 2>&1
cat $O/loads_bench_ab.txt
