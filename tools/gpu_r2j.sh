#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, call J: same-box A/B of (a) the conv K order (old tree = tap-major as of 181fed6; var_tapmajor = the new gather walked
# tap-major, timing only) and (b) the depth up to which split-K partials are folded inside the GEMM
export TMPDIR=/tmp
mkdir -p gpurun_out
B=gligen_amd/build
{
for v in var_old var_tapmajor . var_old .; do
  echo "== $v"
  timeout 200 $B/$v/kbench tools/unet_b8.shapes 10 conv | grep "^conv\|^TOTAL conv" | cut -c1-120
  timeout 200 $B/$v/kbench tools/vae_b4.shapes 5 conv | grep "^TOTAL conv" | cut -c1-120
done
} > gpurun_out/conv_order_ab2.txt 2>&1
grep "==\|TOTAL" gpurun_out/conv_order_ab2.txt
{
for f in 0 2 3 4 8 16 0; do
  echo "== GL_GEMM_SPLIT_FOLD=$f"
  GL_GEMM_SPLIT_FOLD=$f timeout 300 $B/kbench tools/unet_b8.shapes 10 - | grep "^TOTAL gemm\|^TOTAL conv" | cut -c1-160
  GL_GEMM_SPLIT_FOLD=$f timeout 300 python tools/insitu.py 2>/dev/null | head -1
done
echo "== old tree"; timeout 300 $B/var_old/kbench tools/unet_b8.shapes 10 - | grep "^TOTAL" | cut -c1-160
} > gpurun_out/fold_ab2.txt 2>&1
cat gpurun_out/fold_ab2.txt
GL_GEMM_SPLIT_FOLD=0 timeout 300 $B/kbench tools/unet_b8.shapes 10 - > gpurun_out/kbench_fold0.txt 2>&1
GL_GEMM_SPLIT_FOLD=16 timeout 300 $B/kbench tools/unet_b8.shapes 10 - > gpurun_out/kbench_fold16.txt 2>&1
