#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call O: norm kernels three ways on one box -- main (gn_small with the slab in registers, ln by chunk count), var_old (the
# previous commit: ln with three unconditional loads), var_orig (two commits back: guarded loads everywhere)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3o
mkdir -p $O
B=gligen_amd/build
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or layernorm" ) > $O/pytest_norm.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error|^E  " $O/pytest_norm.log | cut -c1-300 | head
{
for round in 1 2; do
for arm in main old orig; do
  k=$B/kbench; [ $arm != main ] && k=$B/var_$arm/kbench
  echo "== $arm, round $round"
  timeout 200 $k tools/unet_b8.shapes 10 gn | grep "^TOTAL gn\|gn 8 256 1280 0 1 6\|gn 8 64 1280 0 1 11\|gn 8 256 1280 1280" | cut -c1-100
  timeout 200 $k tools/unet_b8.shapes 10 ln | grep "^TOTAL ln\|ln 8 4096 0 4096\|ln 8 1024 0 1024\|ln 8 256 0 256" | cut -c1-100
done
done
} > $O/norm_3way.txt 2>&1
cat $O/norm_3way.txt
