#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call C: lean attn2_kernel (branch-free DMA issue, compile-time ring slots): attention op tests + per-shape A/B
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3c
mkdir -p $O
K=$PWD/gligen_amd/build/kbench
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention" ) > $O/pytest_attn.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed|Error" $O/pytest_attn.log | cut -c1-300
for v in 0 1 2; do
  echo "== GL_ATTN_V2=$v"; GL_ATTN_V2=$v timeout 120 $K tools/attn.shapes 20 2>&1 | grep "^attn 8 8 40"
done > $O/attn_kbench.txt 2>&1
cat $O/attn_kbench.txt
