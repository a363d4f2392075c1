#!/bin/bash
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
K=$PWD/gligen_amd/build/kbench
S=$PWD/tools/unet_b8.shapes
OUT=$PWD/gpurun_out/pmc2
run() { # name variant filter counters...
  local name=$1; local var=$2; local filt=$3; shift 3
  ( cd /tmp && GL_GEMM_VARIANT=$var timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $K $S 2 "$filt" ) > $OUT/$name.log 2>&1
}
for var in 2 3; do
for spec in "conv64:conv 8 64 64 320 0 320 1 0" "geglu:gemm 32768 2560 320 1" "conv32:conv 8 32 32 640 0 640 1 0"; do
  n=${spec%%:*}; f=${spec#*:}
  run ${n}_v${var}_a $var "$f" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
  run ${n}_v${var}_b $var "$f" TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
done
done
du -sh $OUT
