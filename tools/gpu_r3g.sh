#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call G: wide GEMM kernel with a three-stage LDS ring (two K tiles in flight) and XCD-contiguous item order, against the
# two-stage / linear-order builds (tools/build_variant.sh w20 / w21 / w30: -DGL_WIDE_NST=2|3 -DGL_WIDE_XCD=0|1), per shape and whole path
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3g
mkdir -p $O
B=gligen_amd/build
mkdir -p $B/var_main && cp gligen_amd/libgligen_amd.so $B/var_main/ && cp $B/kbench $B/var_main/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "ln_folded or linear or geglu or gemm" ) > $O/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_ops.log | cut -c1-300 | head -20
{
for round in 1 2; do
for arm in main w20 w21 w30; do
  echo "== $arm (GEGLU projections: the wide kernel's default problems), round $round"
  timeout 200 $B/var_$arm/kbench tools/unet_b8.shapes 10 gemm check | grep "^gemm [0-9]* [0-9]* [0-9]* 1 \|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-150
done
done
for arm in main w20; do
  echo "== $arm GL_GEMM_WIDE=2 (every eligible problem on the wide kernel)"
  GL_GEMM_WIDE=2 timeout 200 $B/var_$arm/kbench tools/unet_b8.shapes 10 gemm check | grep "256x\|^TOTAL gemm\|CHECK\|MISMATCH\|mismatch" | cut -c1-150
done
} > $O/wide_ring_kbench.txt 2>&1
cat $O/wide_ring_kbench.txt
{
for arm in main w20 main w20; do
  echo "== $arm"
  cp $B/var_$arm/libgligen_amd.so gligen_amd/libgligen_amd.so
  timeout 400 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'eager_sum', r['eager_sum_ms'], 'dominant', r['kernel'], round(r['achieved'],1), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp $B/var_main/libgligen_amd.so gligen_amd/libgligen_amd.so
} > $O/wide_ring_bench_ab.txt 2>&1
cat $O/wide_ring_bench_ab.txt
