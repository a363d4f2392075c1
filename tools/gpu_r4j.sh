#!/bin/bash
# round 4 (j): row-local feed-forward kernel, DMA issue schedule: the next stage's 16 pieces per wave behind every MFMA (ev1), every
# 2nd (ev2), every 3rd (ev3), every 4th (base) of the chunk; chained forms included
export GL_DEV_SWITCHES=1
O=gpurun_out/r4j; mkdir -p $O
for r in 1 2 3; do
  for v in base ev1 ev2 ev4; do
    timeout 120 gligen_amd/build/var_$v/kbench tools/ffn.shapes 5 > $O/$v.$r.txt 2>&1
    echo "== $v run $r rc=$? $(grep '^FFN' $O/$v.$r.txt | sed 's/FFN\(C*\) M\([0-9]*\) C320 \([a-z0-9]*\).*\(fused\|launch\) \([0-9.]*\) us.*maxdiff\( out\)* \([0-9.]*\) .*\(ok\|BAD\|MISMATCH\).*/\1M\2\3 \5 \8;/' | tr '\n' ' ')"
  done
done
