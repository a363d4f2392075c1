#!/bin/bash
# round 4 (l): attn2_kernel with a three-slot K / V^T ring (stage j+2 requested in iteration j, counted vmcnt waits):
# GL_ATTN_V2=3 against the shipped two-slot form (=1), alternated on one box; then the attention op tests under =3
export GL_DEV_SWITCHES=1
O=gpurun_out/r4l; mkdir -p $O
K=gligen_amd/build/kbench
for r in 1 2 3; do
  for v in 1 3; do
    GL_ATTN_V2=$v timeout 120 $K tools/attn.shapes 5 > $O/v$v.$r.txt 2>&1
    echo "== V2=$v run $r rc=$? $(grep '^attn' $O/v$v.$r.txt | awk '{printf "%s/%s/%s: %s  ", $4, $5, $6, $8}')"
  done
done
( GL_ATTN_V2=3 timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "attn or attention" ) > $O/pytest_v3.log 2>&1
tail -5 $O/pytest_v3.log
