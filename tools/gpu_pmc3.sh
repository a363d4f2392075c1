#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# PMC passes (each in its own rocprofv3 run, --kernel-trace only) on single GEMM/conv shapes: v3 vs v5-wide
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc4
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/pmc4
run() { # name force filter counters...
  local name=$1; local force=$2; local filt=$3; shift 3
  ( cd /tmp && GL_GEMM_VARIANT=4 KB_FORCE=$force timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $K $S 2 "$filt" ) > $OUT/$name.log 2>&1
}
for cfg in "w:8,5,1" "p:4,5,1" "q:2,5,1"; do
v=${cfg%%:*}; force=${cfg#*:}
for spec in "conv64:conv 8 64 64 320 0 320 1 0" "ffout:gemm 32768 320 1280 0"; do
  n=${spec%%:*}; f=${spec#*:}
  run ${n}_${v}_a $force "$f" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
  run ${n}_${v}_b $force "$f" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run ${n}_${v}_c $force "$f" SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
done
done
find $OUT -name "*counter_collection.csv" | head -20
du -sh $OUT
