#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call K: conv gather with 24-bit multiplies vs the tap-major tree (181fed6) on one box, op tests
export TMPDIR=/tmp
mkdir -p gpurun_out
B=gligen_amd/build
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q ) > gpurun_out/pytest_ops.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_ops.log | cut -c1-250
{
for v in var_old . var_old . ; do
  echo "== $v"
  timeout 200 $B/$v/kbench tools/unet_b8.shapes 10 conv | grep "^conv\|^TOTAL conv" | cut -c1-120
  timeout 200 $B/$v/kbench tools/vae_b4.shapes 5 conv | grep "^conv\|^TOTAL conv" | cut -c1-120
done
} > gpurun_out/conv_order_ab3.txt 2>&1
grep "==\|TOTAL" gpurun_out/conv_order_ab3.txt
for v in var_old . ; do timeout 300 $B/$v/kbench tools/unet_b8.shapes 10 - | grep "^TOTAL" | cut -c1-100; done
timeout 300 python tools/insitu.py 2>/dev/null | head -1
