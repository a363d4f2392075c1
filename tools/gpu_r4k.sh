#!/bin/bash
# round 4 (k): attn2_kernel (d = 40 self-attention) variants built by tools/build_attn_variant.sh, alternated on one box
#   usage: bash tools/gpu_r4k.sh v1 v2 ...
export GL_DEV_SWITCHES=1
O=gpurun_out/r4k; mkdir -p $O
for r in 1 2 3; do
  for v in "$@"; do
    timeout 120 gligen_amd/build/var_$v/kbench tools/attn1.shapes 5 > $O/$v.$r.txt 2>&1
    echo "== $v run $r rc=$? $(grep '^attn' $O/$v.$r.txt | awk '{printf "%s/%s: %s us  ", $5, $6, $8}')"
  done
done
