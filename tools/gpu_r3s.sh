#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call S: halo conv kernel with the shift-invariant LDS swizzle of its halo buffer (chunk ^= row & 6) against the previous
# commit's library: conv op tests, per-shape check + timing (UNet and VAE), whole path, LDS conflict counters of both
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3s
mkdir -p $O
B=gligen_amd/build
cp gligen_amd/libgligen_amd.so $B/libgligen_amd.main.so
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv" ) > $O/pytest_conv.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_conv.log | cut -c1-300 | head
{
for round in 1 2; do
for arm in main old; do
  k=$B/kbench; [ $arm != main ] && k=$B/var_$arm/kbench
  echo "== $arm, round $round"
  timeout 200 $k tools/unet_b8.shapes 10 conv check | grep "256x\|^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-110
  timeout 200 $k tools/vae_b4.shapes 5 conv check | grep "^TOTAL conv\|CHECK\|MISMATCH\|mismatch" | cut -c1-110
done
done
} > $O/halo_swizzle_kbench.txt 2>&1
cat $O/halo_swizzle_kbench.txt
{
for arm in main old main old; do
  echo "== $arm"
  if [ $arm = old ]; then cp $B/var_old/libgligen_amd.so gligen_amd/libgligen_amd.so; else cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so; fi
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'vae_ms', d.get('vae_decode_ms'), 'sclk', d['gpu_clocks']['sclk_mhz']['mean']); [print('   ', k['kernel'], k['ms'], k.get('TFLOP/s')) for k in r['kernels'] if 'halo' in k['kernel']]"
done
cp $B/libgligen_amd.main.so gligen_amd/libgligen_amd.so
} > $O/halo_swizzle_bench_ab.txt 2>&1
cat $O/halo_swizzle_bench_ab.txt
R=$PWD
for arm in main old; do
  k=$R/$B/kbench; [ $arm != main ] && k=$R/$B/var_$arm/kbench
  ( cd /tmp && timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$arm -- $k $R/tools/halo.shapes 2 conv ) > $O/pmc_$arm.log 2>&1
  python tools/pmc_summarize.py $O/pmc_lds_$arm.csv $O/pmc_$arm > /dev/null; rm -rf $O/pmc_$arm
  echo "== $arm"; grep "conv_halo" $O/pmc_lds_$arm.csv | cut -c1-120
done
