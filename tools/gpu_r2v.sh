#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call V: K step 1 deferred by one tile in the persistent GEMM / conv kernel (all tiles but 128x160) against the same tree
# built with -DGL_GEMM_DEFER_S1=0: kbench per class + check, op tests, then bench.py with the libraries swapped
export TMPDIR=/tmp
mkdir -p gpurun_out
B=gligen_amd/build
L=gligen_amd/libgligen_amd.so
{
for v in var_nodefer . var_nodefer .; do
  echo "== $v"
  timeout 300 $B/$v/kbench tools/unet_b8.shapes 10 - check | grep "^TOTAL\|CHECK\|MISM" | cut -c1-120
done
} > gpurun_out/defer_kbench.txt 2>&1
cat gpurun_out/defer_kbench.txt
timeout 300 $B/var_nodefer/kbench tools/unet_b8.shapes 10 - > gpurun_out/kbench_nodefer.txt 2>&1
timeout 300 $B/kbench tools/unet_b8.shapes 10 - > gpurun_out/kbench_defer.txt 2>&1
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
cp $L /tmp/new.so
{
for arm in nodefer new nodefer new; do
  if [ $arm = nodefer ]; then cp $B/var_nodefer/libgligen_amd.so $L; else cp /tmp/new.so $L; fi
  echo "== $arm"
  timeout 300 python tools/insitu.py 2>/dev/null | head -1
  timeout 400 python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench images/s', round(d['value'],4), 'unet_step_ms', round(d['unet_step_ms'],3), 'vae ms', round(d['vae_decode_ms'],2), 'sclk', d['gpu_clocks']['sclk_mhz']['mean'])"
done
cp /tmp/new.so $L
} > gpurun_out/defer_bench_ab.txt 2>&1
cat gpurun_out/defer_bench_ab.txt
