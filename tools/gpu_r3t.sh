#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 3, call T: LDS bank-conflict counters of the halo kernel with the old and the new halo-buffer swizzle (kbench, halo shapes)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r3t
mkdir -p $O
B=gligen_amd/build
for arm in main old; do
  k=$R/$B/kbench; [ $arm != main ] && k=$R/$B/var_$arm/kbench
  ( cd /tmp && timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$arm -- $k $R/tools/halo.shapes 2 conv ) > $O/pmc_$arm.log 2>&1
  python tools/pmc_summarize.py $O/pmc_lds_$arm.csv $O/pmc_$arm > /dev/null; rm -rf $O/pmc_$arm
  echo "== $arm"; grep "conv_halo" $O/pmc_lds_$arm.csv | cut -c1-120
done
