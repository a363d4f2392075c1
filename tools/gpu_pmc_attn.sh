#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_attn
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/pmc_attn
run() { local name=$1; local filt=$2; shift 2
  ( cd /tmp && timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $K $S 2 "$filt" ) > $OUT/$name.log 2>&1
}
for spec in "a64:attn 8 8 40 4096 4096" "a32:attn 8 8 80 1024 1024"; do
  n=${spec%%:*}; f=${spec#*:}
  run ${n}_a "$f" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
  run ${n}_b "$f" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC
  run ${n}_c "$f" SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INST_LEVEL_LDS
done
