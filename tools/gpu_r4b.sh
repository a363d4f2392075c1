#!/bin/bash
# round 4 (b): the row-local feed-forward kernel: check + timing (product library), then its ablations
# (variant library built by tools/build_ffn_variant.sh ffnabl -DGL_FFN_ABLATE -fno-slp-vectorize)
export GL_DEV_SWITCHES=1
O=gpurun_out/r4b; mkdir -p $O
timeout 300 gligen_amd/build/kbench tools/ffn.shapes 20 > $O/ffn_kbench.txt 2>&1
echo "rc=$?" >> $O/ffn_kbench.txt
grep "FFN\|rc=" $O/ffn_kbench.txt | cut -c1-200
timeout 300 gligen_amd/build/var_ffnabl/kbench tools/ffn2.shapes 20 > $O/ffn_ablation.txt 2>&1
echo "rc=$?" >> $O/ffn_ablation.txt
grep -v "^TOTAL\|^shape" $O/ffn_ablation.txt | cut -c1-200
