#!/bin/bash
# round 4 (m): gemm.hip built under other LLVM scheduling strategies (tools/build_variant.sh NAME -mllvm -amdgpu-sched-strategy=...),
# the UNet's kernel table per variant, alternated on one box
export GL_DEV_SWITCHES=1
O=gpurun_out/r4m; mkdir -p $O
for r in 1 2; do
  for v in "$@"; do
    timeout 200 gligen_amd/build/var_$v/kbench tools/unet_b8.shapes 5 > $O/$v.$r.txt 2>&1
    echo "== $v run $r rc=$? $(grep '^TOTAL' $O/$v.$r.txt | awk '{printf "%s %s ms  ", $2, $3}')"
  done
done
