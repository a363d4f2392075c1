#!/bin/bash
export GL_DEV_SWITCHES=1
O=gpurun_out/r4c; mkdir -p $O
for r in 1 2; do
timeout 300 gligen_amd/build/kbench tools/ffn.shapes 20 > $O/ffn_kbench.$r.txt 2>&1
echo "rc=$?" >> $O/ffn_kbench.$r.txt
grep "FFN\|rc=\|rows with" $O/ffn_kbench.$r.txt | cut -c1-220
done
