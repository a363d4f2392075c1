#!/bin/bash
# PMC passes over a few representative kernels (kbench filters). Usage: gpu_pmc.sh
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/pmc
rocprofv3 -L > $OUT/counters.txt 2>&1
run() { # name filter counters...
  local name=$1; local filt=$2; shift 2
  ( cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $K $S 2 "$filt" ) > $OUT/$name.log 2>&1
}
for spec in "attn40:attn 8 8 40 4096 4096" "geglu:gemm 32768 2560 320 1" "conv64:conv 8 64 64 320 0 320 1 0" "lin320:gemm 32768 320 320 0" "conv16:conv 8 16 16 1280 0 1280 1 0"; do
  n=${spec%%:*}; f=${spec#*:}
  run ${n}_a "$f" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  run ${n}_b "$f" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES
done
find $OUT -name "*.csv" | head -50
du -sh $OUT
