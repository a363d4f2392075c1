#!/bin/bash
# round 4 (i): row-local feed-forward kernel variants on one box (tools/build_ffn_variant.sh): base = the tree (-fno-honor-nans),
# hn = with the canonicalising v_max back, ev2 / ev3 = the next stage's DMA pieces behind every 2nd / 3rd MFMA instead of every
# 4th, rd12 / rd6 = fragment reads 12 / 6 blocks ahead instead of 8; then the op tests on the product library
export GL_DEV_SWITCHES=1
O=gpurun_out/r4i; mkdir -p $O
for r in 1 2 3; do
  for v in base hn ev2 ev3 rd12 rd6; do
    timeout 120 gligen_amd/build/var_$v/kbench tools/ffn2.shapes 5 > $O/$v.$r.txt 2>&1
    echo "== $v run $r rc=$? $(grep '^FFN' $O/$v.$r.txt | sed 's/FFN M\([0-9]*\).*fused \([0-9.]*\) us.*maxdiff \(.*\)/M\1 \2us \3/' | tr '\n' ' ')"
  done
done
timeout 120 gligen_amd/build/var_base/kbench tools/ffn.shapes 5 > $O/base_full.txt 2>&1; grep "^FFN" $O/base_full.txt | cut -c1-170
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "feedforward or ff_chain or geglu" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
