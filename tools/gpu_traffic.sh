#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# HBM-side traffic of the bench command: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (--kernel-trace only; eager
# launches: counter collection over hipGraph replays crashes rocprofv3 here), folded to per-launch means per kernel symbol
# -> gpurun_out/pmc_traffic.csv, and per PROBLEM (symbol + M,N,K via the engine's launch log) -> gpurun_out/pmc_traffic_per_problem.csv
# (both copied to profiles/<round>/). Autotuning is off in these passes so that every pass issues the same launch sequence.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
for c in f:FETCH_SIZE w:WRITE_SIZE; do
  n=${c%%:*}; ctr=${c#*:}
  ( cd /tmp && GL_GEMM_AUTOTUNE=0 GL_FF_POLICY=1 GL_LAUNCH_LOG=$R/gpurun_out/launch_$n.log timeout 700 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -- \
      python $R/bench.py --steps 1 --warmup 0 --lanes 1 --no-graph --plms-steps 2 --no-cpu-baseline --no-train-step --no-ff-ab ) > gpurun_out/pmc_$n.log 2>&1
  tail -1 gpurun_out/pmc_$n.log | cut -c1-300
done
python tools/pmc_summarize.py gpurun_out/pmc_traffic.csv gpurun_out/pmc_f gpurun_out/pmc_w --launch-log gpurun_out/launch_f.log --per-problem gpurun_out/pmc_traffic_per_problem.csv
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
head -12 gpurun_out/pmc_traffic.csv; head -12 gpurun_out/pmc_traffic_per_problem.csv
