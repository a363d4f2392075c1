#!/bin/bash
export GL_DEV_SWITCHES=1
O=gpurun_out/r4d; mkdir -p $O
for v in "$@"; do
  for r in 1 2; do
    timeout 300 gligen_amd/build/var_$v/kbench tools/ffn2.shapes 5 > $O/$v.$r.txt 2>&1
    echo "== $v run $r rc=$?"; grep "FFN\|rows with\|ablation" $O/$v.$r.txt | cut -c1-160
  done
done
