#!/bin/bash
# round 4 (a): first run of the row-local feed-forward kernel: check against the two-GEMM form + per-shape timing
export GL_DEV_SWITCHES=1
O=gpurun_out/r4a; mkdir -p $O
timeout 300 gligen_amd/build/kbench tools/ffn.shapes 20 > $O/ffn_kbench.txt 2>&1
echo "rc=$?" >> $O/ffn_kbench.txt
cat $O/ffn_kbench.txt | cut -c1-200
