#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# MFMA-pipe and LDS counters of the bench command's kernels, one rocprofv3 --pmc pass each (--kernel-trace only; eager launches as in
# gpu_traffic.sh), folded to per-symbol means -> gpurun_out/pmc_mfma.csv (copied to profiles/<round>/). Derived per symbol by
# tools/pmc_mfma_report.py: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) (both summed over the 8 XCDs),
# LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  ( cd /tmp && GL_GEMM_AUTOTUNE=0 GL_FF_POLICY=1 timeout 700 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_m$i -- \
      python $R/bench.py --steps 1 --warmup 0 --lanes 1 --no-graph --plms-steps 2 --no-cpu-baseline --no-train-step --no-ff-ab ) > gpurun_out/pmc_m$i.log 2>&1
  tail -1 gpurun_out/pmc_m$i.log | cut -c1-200
done
python tools/pmc_summarize.py gpurun_out/pmc_mfma.csv gpurun_out/pmc_m1 gpurun_out/pmc_m2
rm -rf gpurun_out/pmc_m1 gpurun_out/pmc_m2
python tools/pmc_mfma_report.py gpurun_out/pmc_mfma.csv | tee gpurun_out/pmc_mfma_report.txt
